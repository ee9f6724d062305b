// dazz_main.cpp -- the rest of the executable boundary of DENTIST's hot path over libdentist_hip.so:
// one multi-call binary, the tool is chosen by argv[0] (or `--tool <name>` as first argument):
//
//   fasta2DB / fasta2DAM  -i <db>  (FASTA on stdin) | <db> <fasta>...   dazzler.d:6233-6330
//   DBsplit [-f] [-a] [-x<n>] [-s<mb>] <db>                             dazzler.d:6332-6345
//   DBrm <db>...                                                        dazzler.d:6115-6119
//   DBdust <db>                      writes the `dust` mask track       processPileUps/package.d:476, 655
//   DBdump [-r -h -s -i] <db> [ids | a-b]   SURVEY Appendix B grammar   dazzler.d:6445-6505, parser :2788-3078
//   DBshow [-n] <db> [ids]           FASTA / scaffold structure lines   dazzler.d:4609-4690, 6507-6517
//   LAmerge <out.las> <in.las>...                                       snakemake/Snakefile:1173-1185
//   DAScover -v <db> <las> ; DASqv -v -c<cov> <db> <las>   `qual` track dazzler.d:6142-6156
//   computeintrinsicqv -d<depth> <db> <las>             `inqual` track  dazzler.d:6172-6183
//   daccord [-t<n>] [-I<i>,<j>] [-f] [--eprofonly] <las> <db>           dazzler.d:6185-6231
//                                    consensus FASTA on stdout; --eprofonly writes <las>.eprof
//   merge-insertions <merged.db> <batch.db>...   (DENTIST's own sub-command, commands/mergeInsertions.d:42-164,
//                                    snakemake/Snakefile:1315-1334; here so that batches written by this library
//                                    can be merged without the D binary)
//   LAsplit <target with @ or #> <parts> < <source.las>     the workflow's split of a merged .las for the validation blocks
//                                    (snakemake/Snakefile:1426-1434): nearly equal parts, cut between A reads
//   Catrack [-v] [-f] [-d] <db> <track>          block mask tracks .<db>.<block>.<track>.{anno,data} concatenated into the
//                                    DB's track (snakemake/Snakefile:1111-1123; track layout dazzler.d:4943-5170)
//   TANmask [-v] [-l<int(500)>] [-n<track(tan)>] <db> <TAN las>...   self alignments of a read -> mask intervals
//                                    (snakemake/Snakefile:1095-1108); the block's track when the .las names a block
// DENTIST only sees exit codes, files and stdout of these tools; flags it never emits are rejected.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/dentist_hip.h"

static std::string g_tool;
static void die(const std::string &msg, int rc = 1)
{
    fprintf(stderr, "%s: %s\n", g_tool.c_str(), msg.c_str());
    exit(rc);
}
#define CHK(call)                                                                                 \
    do {                                                                                          \
        if (int rc_ = (call)) die(std::string(#call) + ": " + dh_last_error(), rc_ < 0 ? -rc_ : rc_); \
    } while (0)

static std::string slurp(FILE *f)
{
    std::string s;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, got);
    return s;
}
static bool is_dam(const std::string &p) { return p.size() > 4 && p.compare(p.size() - 4, 4, ".dam") == 0; }

// ---------------------------------------------------------------------------------- DB tools
static int tool_fasta2(bool dam, const std::vector<std::string> &args)
{
    bool from_stdin = false;
    std::vector<std::string> pos;
    for (const std::string &a : args) {
        if (a == "-i" || a.compare(0, 2, "-i") == 0)
            from_stdin = true;
        else if (a == "-v")
            ;
        else if (a[0] == '-')
            die("unknown option " + a);
        else
            pos.push_back(a);
    }
    if (pos.empty()) die("usage: fasta2DB|fasta2DAM [-v] <path> ( -i | <input:fasta> ... )");
    std::string text;
    if (from_stdin)
        text = slurp(stdin);
    else
        for (size_t i = 1; i < pos.size(); i++) {
            FILE *f = fopen(pos[i].c_str(), "r");
            if (!f) die("cannot open " + pos[i]);
            text += slurp(f);
            fclose(f);
            if (!text.empty() && text.back() != '\n') text += '\n';
        }
    if (dam)
        CHK(dh_dazz_create_dam(pos[0].c_str(), text.data(), (int64_t)text.size()));
    else
        CHK(dh_dazz_create_db(pos[0].c_str(), text.data(), (int64_t)text.size()));
    return 0;
}

static int tool_dbsplit(const std::vector<std::string> &args)
{
    int cutoff = 0, all = 0;
    long long size = 200;
    std::string db;
    for (const std::string &a : args) {
        if (a == "-a")
            all = 1;
        else if (a == "-f")
            ;
        else if (a.compare(0, 2, "-x") == 0)
            cutoff = atoi(a.c_str() + 2);
        else if (a.compare(0, 2, "-s") == 0)
            size = (long long)atof(a.c_str() + 2);
        else if (a[0] == '-')
            die("unknown option " + a);
        else
            db = a;
    }
    if (db.empty()) die("usage: DBsplit [-af] [-x<int>] [-s<double(200.)>] <path:db|dam>");
    CHK(dh_dazz_split(db.c_str(), cutoff, all, size));
    return 0;
}

static int tool_dbrm(const std::vector<std::string> &args)
{
    for (const std::string &a : args)
        if (a[0] != '-') CHK(dh_dazz_remove(a.c_str()));
    return 0;
}

static dh_dazz *open_dazz(const std::string &path)
{
    dh_dazz *d = nullptr;
    CHK(dh_dazz_open(path.c_str(), &d));
    return d;
}

// record numbers (1-based) from `ids` / `a-b` arguments; empty = all
static std::vector<int32_t> record_list(const std::vector<std::string> &sel, int32_t first, int32_t n)
{
    std::vector<int32_t> ids;
    if (sel.empty()) {
        for (int32_t i = 0; i < n; i++) ids.push_back(i);
        return ids;
    }
    for (const std::string &s : sel) {
        int a = 0, b = 0;
        if (sscanf(s.c_str(), "%d-%d", &a, &b) == 2)
            ;
        else if (sscanf(s.c_str(), "%d", &a) == 1)
            b = a;
        else
            die("bad record selector " + s);
        for (int x = a; x <= b; x++) {
            const int32_t loc = x - 1 - first;
            if (loc < 0 || loc >= n) die("record " + std::to_string(x) + " is not in the DB");
            ids.push_back(loc);
        }
    }
    return ids;
}

static char qv_char(int q) { return q < 26 ? (char)('a' + q) : (char)('A' + std::min(q, 50) - 26); }

static int tool_dbdump(const std::vector<std::string> &args)
{
    bool fr = false, fh = false, fs = false, fi = false;
    std::string db;
    std::vector<std::string> sel;
    for (const std::string &a : args) {
        if (a[0] == '-' && a.size() > 1 && !isdigit((unsigned char)a[1])) {
            for (size_t x = 1; x < a.size(); x++) switch (a[x]) {
                case 'r': fr = true; break;
                case 'h': fh = true; break;
                case 's': fs = true; break;
                case 'i': fi = true; break;
                case 'u': case 'U': break;
                default: die("unknown option " + a);
                }
        } else if (db.empty())
            db = a;
        else
            sel.push_back(a);
    }
    if (db.empty()) die("usage: DBdump [-rhsi] <path:db|dam> [ <reads:range> ... ]");
    dh_dazz *d = open_dazz(db);
    const int32_t n = dh_dazz_nreads(d), first = dh_dazz_first_id(d);
    const std::vector<int32_t> ids = record_list(sel, first, n);
    const int64_t *off = dh_dazz_offsets(d);
    const uint8_t *bases = dh_dazz_bases(d);
    std::vector<int64_t> qptr;
    std::vector<uint8_t> qv;
    if (fi) {
        qptr.resize((size_t)n + 1);
        const int64_t m = dh_dazz_read_track(d, db.c_str(), "qual", qptr.data(), nullptr, 0);
        if (m < 0) die(std::string("-i needs the qual track (run DASqv): ") + dh_last_error());
        qv.resize((size_t)std::max<int64_t>(m, 1));
        dh_dazz_read_track(d, db.c_str(), "qual", qptr.data(), qv.data(), m);
    }
    // header lines: totals and maxima of every line type (dazzler.d:2788-2813 reads `+ R` and `+ S`)
    int64_t tot_s = 0, max_s = 0, tot_h = 0, max_h = 0, tot_i = 0, max_i = 0;
    for (int32_t i : ids) {
        const int64_t l = off[i + 1] - off[i];
        tot_s += l;
        max_s = std::max(max_s, l);
        const int64_t hl = (int64_t)strlen(dh_dazz_header(d, i));
        tot_h += hl;
        max_h = std::max(max_h, hl);
        if (fi) {
            const int64_t ql = qptr[(size_t)i + 1] - qptr[(size_t)i];
            tot_i += ql;
            max_i = std::max(max_i, ql);
        }
    }
    printf("+ R %zu\n+ M 0\n", ids.size());
    if (fh) printf("+ H %lld\n@ H %lld\n", (long long)tot_h, (long long)max_h);
    if (fs) printf("+ S %lld\n@ S %lld\n", (long long)tot_s, (long long)max_s);
    if (fi) printf("+ I %lld\n@ I %lld\n", (long long)tot_i, (long long)max_i);
    static const char ACGT[] = "acgtn";
    const bool dam = is_dam(db) || (db.find(".db") == std::string::npos && dh_dazz_header(d, 0)[0] == '>');
    std::string seq;
    for (int32_t i : ids) {
        const int64_t l = off[i + 1] - off[i];
        if (fr) printf("R %d\n", first + i + 1);
        if (fh) {
            const char *h = dh_dazz_header(d, i);
            printf("H %zu %s\n", strlen(h), h);
            // L <well | contig in scaffold> <begin> <end> (dazzler.d:1559-1561: length = end - begin)
            printf("L %d %d %lld\n", dh_dazz_origin(d)[i], dh_dazz_fpulse(d)[i], (long long)(dh_dazz_fpulse(d)[i] + l));
            if (!dam) printf("Q 0.%03d\n", dh_dazz_flags(d)[i] & 0x3ff);
        }
        if (fs) {
            seq.resize((size_t)l);
            for (int64_t x = 0; x < l; x++) seq[(size_t)x] = ACGT[std::min<int>(bases[off[i] + x], 4)];
            printf("S %lld %s\n", (long long)l, seq.c_str());
        }
        if (fi) {
            const int64_t a = qptr[(size_t)i], b = qptr[(size_t)i + 1];
            std::string q;
            for (int64_t x = a; x < b; x++) q.push_back(qv_char(qv[(size_t)x]));
            printf("I %lld %s\n", (long long)(b - a), q.c_str());
        }
    }
    dh_dazz_close(d);
    return 0;
}

static int tool_dbshow(const std::vector<std::string> &args)
{
    bool names = false;
    int width = 80;
    std::string db;
    std::vector<std::string> sel;
    for (const std::string &a : args) {
        if (a == "-n")
            names = true;
        else if (a.compare(0, 2, "-w") == 0)
            width = std::max(1, atoi(a.c_str() + 2));
        else if (a == "-u" || a == "-U" || a == "-q")
            ;
        else if (a[0] == '-' && !isdigit((unsigned char)a[1]))
            die("unknown option " + a);
        else if (db.empty())
            db = a;
        else
            sel.push_back(a);
    }
    if (db.empty()) die("usage: DBshow [-n] [-w<int(80)>] <path:db|dam> [ <reads:range> ... ]");
    dh_dazz *d = open_dazz(db);
    const int32_t n = dh_dazz_nreads(d), first = dh_dazz_first_id(d);
    const std::vector<int32_t> ids = record_list(sel, first, n);
    const int64_t *off = dh_dazz_offsets(d);
    const uint8_t *bases = dh_dazz_bases(d);
    const bool dam = is_dam(db) || (n > 0 && dh_dazz_header(d, 0)[0] == '>');
    static const char ACGT[] = "acgtn";
    for (int32_t i : ids) {
        const int64_t l = off[i + 1] - off[i];
        const char *h = dh_dazz_header(d, i);
        const int32_t org = dh_dazz_origin(d)[i], fp = dh_dazz_fpulse(d)[i];
        std::string head;
        if (dam) {  // `<fasta header> :: Contig <idx>[<begin>,<end>]`, dazzler.d:4689-4690
            head = std::string(h[0] == '>' ? "" : ">") + h + " :: Contig " + std::to_string(org) + "[" + std::to_string(fp) + "," +
                   std::to_string(fp + l) + "]";
        } else  // PacBio style: >prolog/well/beg_end RQ=0.xxx (dazzler.d:1389-1393)
            head = ">" + std::string(h) + "/" + std::to_string(org) + "/" + std::to_string(fp) + "_" + std::to_string(fp + l) + " RQ=0.850";
        printf("%s\n", head.c_str());
        if (names) continue;
        for (int64_t x = 0; x < l; x += width) {
            const int64_t e = std::min<int64_t>(l, x + width);
            for (int64_t y = x; y < e; y++) putchar(ACGT[std::min<int>(bases[off[i] + y], 4)]);
            putchar('\n');
        }
    }
    dh_dazz_close(d);
    return 0;
}

// ---------------------------------------------------------------------------------- device tools
struct Dev {
    dh_ctx *ctx = nullptr;
    dh_dazz *dz = nullptr;
    dh_db *db = nullptr;
};
static Dev open_dev(const std::string &path)
{
    Dev v;
    CHK(dh_ctx_create(0, nullptr, &v.ctx));
    v.dz = open_dazz(path);
    CHK(dh_db_create(v.ctx, dh_dazz_bases(v.dz), dh_dazz_offsets(v.dz), dh_dazz_nreads(v.dz), nullptr, &v.db));
    return v;
}
static void close_dev(Dev &v)
{
    dh_db_destroy(v.db);
    dh_dazz_close(v.dz);
    dh_ctx_destroy(v.ctx);
}

static int tool_dbdust(const std::vector<std::string> &args)
{
    std::string db;
    for (const std::string &a : args) {
        if (a[0] == '-') {
            if (strchr("wtmb", a[1]) == nullptr) die("unknown option " + a);
            if ((a[1] == 'w' && atoi(a.c_str() + 2) != 64) || (a[1] == 't' && fabs(atof(a.c_str() + 2) - 2.0) > 1e-9) ||
                (a[1] == 'm' && atoi(a.c_str() + 2) != 10))
                die("only the defaults -w64 -t2.0 -m10 are implemented");
        } else
            db = a;
    }
    if (db.empty()) die("usage: DBdust [-w<int(64)>] [-t<double(2.)>] [-m<int(10)>] <path:db|dam>");
    Dev v = open_dev(db);
    CHK(dh_db_dust(v.db));
    const int32_t n = dh_dazz_nreads(v.dz);
    std::vector<int64_t> ptr((size_t)n + 1);
    const int64_t m = dh_db_get_mask(v.db, ptr.data(), nullptr, 0);
    if (m < 0) die(dh_last_error());
    std::vector<int32_t> iv((size_t)std::max<int64_t>(2 * m, 2));
    dh_db_get_mask(v.db, ptr.data(), iv.data(), m);
    CHK(dh_dazz_write_mask(db.c_str(), "dust", n, ptr.data(), iv.data()));
    close_dev(v);
    return 0;
}

// overlaps of a pile-up DB: ids in the file are trimmed DB ids
static dh_la_set *read_las(const std::string &path, std::vector<dh_la> &las, int32_t first)
{
    dh_la_set *set = nullptr;
    CHK(dh_las_read(path.c_str(), &set));
    const int64_t n = dh_la_set_count(set);
    las.assign(dh_la_set_records(set), dh_la_set_records(set) + n);
    for (dh_la &l : las) {
        l.aread -= first;
        l.bread -= first;
    }
    return set;
}

// DAScover + DASqv / computeintrinsicqv: intrinsic QV per trace tile of every read -> byte track
static int tool_qv(const char *track, const std::vector<std::string> &args, bool write)
{
    int cov = 0;
    std::vector<std::string> pos;
    for (const std::string &a : args) {
        if (a == "-v")
            ;
        else if (a.compare(0, 2, "-c") == 0 || a.compare(0, 2, "-d") == 0)
            cov = atoi(a.c_str() + 2);
        else if (a.compare(0, 2, "-m") == 0 || a.compare(0, 2, "-H") == 0)
            ;
        else if (a[0] == '-')
            die("unknown option " + a);
        else
            pos.push_back(a);
    }
    if (pos.size() != 2) die(std::string("usage: ") + g_tool + " [-v] [-c<int>|-d<int>] <db> <las>");
    if (!write) return 0;  // DAScover: the coverage estimate is folded into DASqv's -c here
    Dev v = open_dev(pos[0]);
    std::vector<dh_la> las;
    dh_la_set *set = read_las(pos[1], las, dh_dazz_first_id(v.dz));
    const int32_t n = dh_dazz_nreads(v.dz), ts = dh_la_set_tspace(set);
    if (cov <= 0) cov = std::max(4, n);
    const int64_t *off = dh_dazz_offsets(v.dz);
    int32_t maxtiles = 1;
    for (int32_t i = 0; i < n; i++) maxtiles = std::max<int32_t>(maxtiles, (int32_t)((off[i + 1] - off[i] + ts - 1) / ts));
    std::vector<uint8_t> qv((size_t)n * maxtiles, 255);
    CHK(dh_tile_qv(v.ctx, v.db, las.data(), (int64_t)las.size(), dh_la_set_trace(set), ts, cov, qv.data(), maxtiles));
    std::vector<int64_t> ptr((size_t)n + 1, 0);
    std::vector<uint8_t> data;
    for (int32_t i = 0; i < n; i++) {
        const int32_t nt = (int32_t)((off[i + 1] - off[i] + ts - 1) / ts);
        for (int32_t t = 0; t < nt; t++) data.push_back(std::min<uint8_t>(qv[(size_t)i * maxtiles + t], 50));
        ptr[(size_t)i + 1] = (int64_t)data.size();
    }
    data.push_back(0);
    CHK(dh_dazz_write_track(pos[0].c_str(), track, n, ptr.data(), data.data()));
    dh_la_set_destroy(set);
    close_dev(v);
    return 0;
}

static int tool_daccord(const std::vector<std::string> &args)
{
    int i0 = -1, i1 = -1, rounds = 3;
    bool eprof_only = false;
    std::vector<std::string> pos;
    for (const std::string &a : args) {
        if (a.compare(0, 2, "-I") == 0) {
            if (sscanf(a.c_str() + 2, "%d,%d", &i0, &i1) != 2) die("bad -I");
        } else if (a == "--eprofonly")
            eprof_only = true;
        else if (a == "-f" || a.compare(0, 2, "-t") == 0 || a.compare(0, 2, "-w") == 0 || a.compare(0, 2, "-a") == 0 ||
                 a.compare(0, 2, "-k") == 0 || a.compare(0, 2, "-m") == 0 || a.compare(0, 2, "-d") == 0 || a.compare(0, 2, "-V") == 0)
            ;
        else if (a.compare(0, 9, "--rounds=") == 0)
            rounds = atoi(a.c_str() + 9);
        else if (a[0] == '-')
            die("unknown option " + a);
        else
            pos.push_back(a);
    }
    if (pos.size() != 2) die("usage: daccord [-t<n>] [-I<i>,<j>] [-f] [--eprofonly] <las> <db>");
    if (eprof_only) {  // the error profile pass: this consensus needs none, the file marks it as done
        FILE *f = fopen((pos[0] + ".eprof").c_str(), "wb");
        if (!f) die("cannot write " + pos[0] + ".eprof");
        fputs("dentist-hip: no error profile needed\n", f);
        fclose(f);
        return 0;
    }
    Dev v = open_dev(pos[1]);
    std::vector<dh_la> las;
    dh_la_set *set = read_las(pos[0], las, dh_dazz_first_id(v.dz));
    const int32_t n = dh_dazz_nreads(v.dz), ts = dh_la_set_tspace(set);
    if (i0 < 0) {
        i0 = 0;
        i1 = n - 1;
    }
    if (i0 > i1 || i1 >= n) die("-I outside the DB");
    const int64_t *off = dh_dazz_offsets(v.dz);
    static const char ACGT[] = "acgtn";
    for (int32_t r = i0; r <= i1; r++) {
        std::vector<uint8_t> out((size_t)(off[r + 1] - off[r]) * 6 + 64);
        int64_t len = 0;
        CHK(dh_consensus(v.ctx, v.db, las.data(), (int64_t)las.size(), dh_la_set_trace(set), ts, r, rounds, out.data(),
                         (int64_t)out.size(), &len));
        // header in daccord's style: read id (0-based) / segment / 0_length
        printf(">%d/0/0_%lld A=[0,%lld]\n", r, (long long)len, (long long)len);
        for (int64_t x = 0; x < len; x += 80) {
            for (int64_t y = x; y < std::min(len, x + 80); y++) putchar(ACGT[std::min<int>(out[(size_t)y], 4)]);
            putchar('\n');
        }
    }
    dh_la_set_destroy(set);
    close_dev(v);
    return 0;
}

static int tool_lamerge(const std::vector<std::string> &args)
{
    std::vector<const char *> in;
    std::string out;
    for (const std::string &a : args) {
        if (a == "-v" || a == "-a" || a.compare(0, 2, "-P") == 0)
            continue;
        if (a[0] == '-') die("unknown option " + a);
        if (out.empty())
            out = a;
        else
            in.push_back(a.c_str());
    }
    if (out.empty() || in.empty()) die("usage: LAmerge [-va] <merge:las> <parts:las> ...");
    if (out.size() < 4 || out.compare(out.size() - 4, 4, ".las") != 0) out += ".las";
    CHK(dh_las_merge(in.data(), (int32_t)in.size(), out.c_str()));
    return 0;
}

// ---------------------------------------------------------------------------------- workflow helpers (host only)
// LAsplit: the records of a .las on stdin in <parts> files of nearly equal size; a file ends only where the A read changes
// (the piles of an A read stay together: every consumer of a block file reads piles).  '@' or '#' in the target is the
// part number, 1-based.
static int tool_lasplit(const std::vector<std::string> &args)
{
    std::vector<std::string> pos;
    for (const std::string &a : args) {
        if (a == "-v") continue;
        if (a[0] == '-' && a.size() > 1) die("unknown option " + a);
        pos.push_back(a);
    }
    if (pos.size() != 2) die("usage: LAsplit <target:path with @> <parts:int> < <source>.las");
    const int32_t parts = atoi(pos[1].c_str());
    const size_t mark = pos[0].find_first_of("@#");
    if (parts < 1 || mark == std::string::npos) die("LAsplit: the target needs a '@' (or '#') and parts >= 1 (a DB as the second argument is not supported)");
    // the codec works on files: stdin goes through a temporary one next to the first target
    std::string tmp = pos[0];
    tmp.replace(mark, 1, "stdin-tmp");
    if (tmp.size() < 4 || tmp.compare(tmp.size() - 4, 4, ".las") != 0) tmp += ".las";
    {
        const std::string raw = slurp(stdin);
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(raw.data(), 1, raw.size(), f) != raw.size() || fclose(f) != 0) die("cannot write " + tmp);
    }
    dh_la_set *set = nullptr;
    const int rc = dh_las_read(tmp.c_str(), &set);
    remove(tmp.c_str());
    if (rc) die(std::string("dh_las_read: ") + dh_last_error());
    const int64_t n = dh_la_set_count(set);
    const dh_la *la = dh_la_set_records(set);
    const uint16_t *tr = dh_la_set_trace(set);
    const int32_t ts = dh_la_set_tspace(set);
    int64_t at = 0;
    for (int32_t k = 1; k <= parts; k++) {
        int64_t end = k == parts ? n : std::min<int64_t>(n, (n * k + parts - 1) / parts);
        end = std::max(end, at);
        while (end > at && end < n && la[end].aread == la[end - 1].aread) end++;  // do not cut a pile
        std::string out = pos[0];
        out.replace(mark, 1, std::to_string(k));
        if (out.size() < 4 || out.compare(out.size() - 4, 4, ".las") != 0) out += ".las";
        // (records keep their trace offsets into the whole trace array)
        CHK(dh_las_write(out.c_str(), la + at, end - at, tr, ts));
        at = end;
    }
    dh_la_set_destroy(set);
    return 0;
}

struct DbPath {
    std::string dir, root, ext;  // dir/root.ext (ext "db" or "dam"), block > 0 when the path names one
    int32_t block = 0;
};
static DbPath parse_db_path(std::string p)
{
    DbPath d;
    const size_t sl = p.find_last_of('/');
    d.dir = sl == std::string::npos ? "." : p.substr(0, sl);
    std::string base = sl == std::string::npos ? p : p.substr(sl + 1);
    for (const char *e : {".dam", ".db"})
        if (base.size() > strlen(e) && base.compare(base.size() - strlen(e), strlen(e), e) == 0) base.resize(base.size() - strlen(e));
    const size_t dot = base.find_last_of('.');
    if (dot != std::string::npos && dot + 1 < base.size() && base.find_first_not_of("0123456789", dot + 1) == std::string::npos) {
        d.block = atoi(base.c_str() + dot + 1);
        base.resize(dot);
    }
    d.root = base;
    for (const char *e : {"dam", "db"}) {
        FILE *f = fopen((d.dir + "/" + d.root + "." + e).c_str(), "r");
        if (f) {
            fclose(f);
            d.ext = e;
            break;
        }
    }
    if (d.ext.empty()) die("DAZZ_DB not found: " + p);
    return d;
}
static int32_t stub_blocks(const DbPath &d)
{
    FILE *f = fopen((d.dir + "/" + d.root + "." + d.ext).c_str(), "r");
    if (!f) die("cannot open the DB stub");
    char line[512];
    int32_t nb = 0;
    while (fgets(line, sizeof(line), f))
        if (sscanf(line, "blocks = %d", &nb) == 1) break;
    fclose(f);
    return nb;
}
static std::string track_file(const DbPath &d, int32_t block, const std::string &name, const char *ext)
{
    return d.dir + "/." + d.root + (block > 0 ? "." + std::to_string(block) : "") + "." + name + "." + ext;
}
static void write_mask_files(const DbPath &d, int32_t block, const std::string &name, int32_t nreads, const std::vector<int64_t> &ptr,
                             const std::vector<int32_t> &iv)
{
    FILE *an = fopen(track_file(d, block, name, "anno").c_str(), "wb"), *da = fopen(track_file(d, block, name, "data").c_str(), "wb");
    if (!an || !da) die("cannot create the track files of " + name);
    const int32_t head[2] = {nreads, 0};  // size 0 marks a mask track (dazzler.d:5143)
    bool ok = fwrite(head, 4, 2, an) == 2;
    for (int32_t i = 0; i <= nreads; i++) {
        const int64_t off = ptr[(size_t)i] * 2 * (int64_t)sizeof(int32_t);
        ok = ok && fwrite(&off, 8, 1, an) == 1;
    }
    if (!iv.empty()) ok = ok && fwrite(iv.data(), 4, iv.size(), da) == iv.size();
    ok = (fclose(an) == 0) && ok;
    ok = (fclose(da) == 0) && ok;
    if (!ok) die("short write of the track files of " + name);
}

// Catrack: the block tracks of a mask concatenated into the track of the whole DB
static int tool_catrack(const std::vector<std::string> &args)
{
    std::vector<std::string> pos;
    bool del = false;
    for (const std::string &a : args) {
        if (a == "-v" || a == "-f") continue;
        if (a == "-d") {
            del = true;
            continue;
        }
        if (a[0] == '-') die("unknown option " + a);
        pos.push_back(a);
    }
    if (pos.size() != 2) die("usage: Catrack [-vfd] <path:db|dam> <track:name>");
    const DbPath d = parse_db_path(pos[0]);
    const int32_t nb = stub_blocks(d);
    if (nb < 1) die("Catrack: the DB has not been split (DBsplit)");
    std::vector<int64_t> ptr{0};
    std::vector<int32_t> iv;
    int32_t nreads = 0;
    for (int32_t b = 1; b <= nb; b++) {
        FILE *an = fopen(track_file(d, b, pos[1], "anno").c_str(), "rb"), *da = fopen(track_file(d, b, pos[1], "data").c_str(), "rb");
        if (!an || !da) die("Catrack: track " + pos[1] + " of block " + std::to_string(b) + " is missing");
        int32_t head[2];
        if (fread(head, 4, 2, an) != 2 || head[1] != 0 || head[0] < 0) die("Catrack: not a mask track (block " + std::to_string(b) + ")");
        std::vector<int64_t> offs((size_t)head[0] + 1);
        if (fread(offs.data(), 8, offs.size(), an) != offs.size()) die("Catrack: truncated .anno of block " + std::to_string(b));
        fclose(an);
        std::vector<int32_t> data;
        int32_t buf[4096];
        size_t got;
        while ((got = fread(buf, 4, 4096, da)) > 0) data.insert(data.end(), buf, buf + got);
        fclose(da);
        if (offs[0] != 0 || offs.back() != (int64_t)data.size() * 4) die("Catrack: .anno and .data of block " + std::to_string(b) + " disagree");
        const int64_t base = ptr.back();
        for (int32_t i = 1; i <= head[0]; i++) {
            if (offs[(size_t)i] < offs[(size_t)i - 1] || offs[(size_t)i] % 8) die("Catrack: corrupted offsets");
            ptr.push_back(base + offs[(size_t)i] / 8);
        }
        iv.insert(iv.end(), data.begin(), data.end());
        nreads += head[0];
    }
    write_mask_files(d, 0, pos[1], nreads, ptr, iv);
    if (del)
        for (int32_t b = 1; b <= nb; b++) {
            remove(track_file(d, b, pos[1], "anno").c_str());
            remove(track_file(d, b, pos[1], "data").c_str());
        }
    return 0;
}

// TANmask: a local alignment of a read with itself (datander) marks a tandem repeat -- the union of its A and B intervals,
// when at least -l long, goes into the mask; intervals of a read are merged.  One track per .las: the block's when the
// file name carries a block number (TAN.<db>.<block>.las), the DB's otherwise.
static int tool_tanmask(const std::vector<std::string> &args)
{
    std::vector<std::string> pos;
    int32_t minlen = 500;
    std::string name = "tan";
    for (const std::string &a : args) {
        if (a == "-v") continue;
        if (a.compare(0, 2, "-l") == 0 && a.size() > 2)
            minlen = atoi(a.c_str() + 2);
        else if (a.compare(0, 2, "-n") == 0 && a.size() > 2)
            name = a.substr(2);
        else if (a[0] == '-')
            die("unknown option " + a);
        else
            pos.push_back(a);
    }
    if (pos.size() < 2) die("usage: TANmask [-v] [-l<int(500)>] [-n<track(tan)>] <subject:db|dam> <overlaps:las> ...");
    const DbPath d = parse_db_path(pos[0]);
    for (size_t f = 1; f < pos.size(); f++) {
        std::string lp = pos[f];
        if (lp.size() < 4 || lp.compare(lp.size() - 4, 4, ".las") != 0) lp += ".las";
        // block of the file: "<...>.<root>.<block>.las"
        int32_t block = 0;
        {
            const std::string stem = lp.substr(0, lp.size() - 4);
            const size_t dot = stem.find_last_of('.');
            if (dot != std::string::npos && dot + 1 < stem.size() && stem.find_first_not_of("0123456789", dot + 1) == std::string::npos)
                block = atoi(stem.c_str() + dot + 1);
        }
        dh_dazz *db = open_dazz(d.dir + "/" + d.root + (block > 0 ? "." + std::to_string(block) : "") + "." + d.ext);
        const int32_t n = dh_dazz_nreads(db), first = dh_dazz_first_id(db);
        dh_la_set *set = nullptr;
        CHK(dh_las_read(lp.c_str(), &set));
        const int64_t nl = dh_la_set_count(set);
        const dh_la *la = dh_la_set_records(set);
        std::vector<std::vector<std::pair<int32_t, int32_t>>> per((size_t)n);
        for (int64_t i = 0; i < nl; i++) {
            if (la[i].aread != la[i].bread || (la[i].flags & DH_FLAG_COMP)) continue;
            const int32_t r = la[i].aread - first;
            if (r < 0 || r >= n) die("TANmask: read id outside the DB block");
            const int32_t b = std::min(la[i].abpos, la[i].bbpos), e = std::max(la[i].aepos, la[i].bepos);
            if (e - b >= minlen) per[(size_t)r].emplace_back(b, e);
        }
        std::vector<int64_t> ptr{0};
        std::vector<int32_t> iv;
        for (int32_t r = 0; r < n; r++) {
            auto &v = per[(size_t)r];
            std::sort(v.begin(), v.end());
            size_t at = iv.size();
            for (const auto &x : v) {
                if (iv.size() > at && x.first <= iv.back())
                    iv.back() = std::max(iv.back(), x.second);
                else {
                    iv.push_back(x.first);
                    iv.push_back(x.second);
                }
            }
            ptr.push_back((int64_t)iv.size() / 2);
        }
        write_mask_files(d, block, name, n, ptr, iv);
        dh_la_set_destroy(set);
        dh_dazz_close(db);
    }
    return 0;
}

static int tool_merge_insertions(const std::vector<std::string> &args)
{
    std::vector<const char *> in;
    std::string out;
    for (const std::string &a : args) {
        if (a == "-v" || a == "-vv" || a == "-vvv") continue;
        if (a[0] == '-') die("unknown option " + a);
        if (out.empty())
            out = a;
        else
            in.push_back(a.c_str());
    }
    if (out.empty() || in.empty()) die("usage: merge-insertions <merged-insertions:db> <insertions:db> ...");
    int64_t n = 0;
    CHK(dh_insertiondb_merge(in.data(), (int32_t)in.size(), out.c_str(), &n));
    fprintf(stderr, "{\"numInputFiles\":%zu,\"totalNumInsertions\":%lld}\n", in.size(), (long long)n);
    return 0;
}

int main(int argc, char **argv)
{
    g_tool = argv[0];
    const size_t slash = g_tool.find_last_of('/');
    if (slash != std::string::npos) g_tool = g_tool.substr(slash + 1);
    int first = 1;
    if (argc > 2 && std::string(argv[1]) == "--tool") {
        g_tool = argv[2];
        first = 3;
    }
    std::vector<std::string> args(argv + first, argv + argc);
    if (g_tool == "fasta2DB") return tool_fasta2(false, args);
    if (g_tool == "fasta2DAM") return tool_fasta2(true, args);
    if (g_tool == "DBsplit") return tool_dbsplit(args);
    if (g_tool == "DBrm") return tool_dbrm(args);
    if (g_tool == "DBdump") return tool_dbdump(args);
    if (g_tool == "DBshow") return tool_dbshow(args);
    if (g_tool == "DBdust") return tool_dbdust(args);
    if (g_tool == "LAmerge") return tool_lamerge(args);
    if (g_tool == "DAScover") return tool_qv("qual", args, false);
    if (g_tool == "DASqv") return tool_qv("qual", args, true);
    if (g_tool == "computeintrinsicqv") return tool_qv("inqual", args, true);
    if (g_tool == "daccord") return tool_daccord(args);
    if (g_tool == "merge-insertions") return tool_merge_insertions(args);
    if (g_tool == "LAsplit") return tool_lasplit(args);
    if (g_tool == "Catrack") return tool_catrack(args);
    if (g_tool == "TANmask") return tool_tanmask(args);
    die("unknown tool (expected fasta2DB fasta2DAM DBsplit DBrm DBdump DBshow DBdust LAmerge DAScover DASqv "
        "computeintrinsicqv daccord merge-insertions)");
    return 1;
}
