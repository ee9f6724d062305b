// aligner_main.cpp -- drop-in `daligner` / `damapper` / `datander` executables over libdentist_hip.so.
//
// DENTIST spawns these tools by name with cwd = output directory and absolute (or stub) DB
// arguments and then expects `<A>.<B>.las` next to it (source/dentist/dazzler.d:6121-6170,
// getLasFile :4339-4354; literal instance tests/test-commands.sh:190-197).  The flag subset DENTIST
// emits is parsed (source/dentist/commandline.d:2886-2955); -m<track> loads the
// mask files DENTIST writes (dazzler.d:4870-5170) and excludes masked k-mers from seeding.  The mode is chosen
// by argv[0] (daligner | damapper | datander) or `--mode`.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/dentist_hip.h"

static void die(const std::string &msg, int rc = 1)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(rc);
}
#define CHK(call)                                                                                 \
    do {                                                                                          \
        if (int rc_ = (call)) die(std::string(#call) + ": " + dh_last_error(), rc_ < 0 ? -rc_ : rc_); \
    } while (0)

struct OpenDb {
    dh_dazz *dz = nullptr;
    dh_db *dev = nullptr;
    std::string name;  // root[.block] as used in .las file names
};

static std::string las_name_part(const std::string &arg)
{
    std::string s = arg;
    const size_t slash = s.find_last_of('/');
    if (slash != std::string::npos) s = s.substr(slash + 1);
    for (const char *e : {".dam", ".db"}) {
        const size_t n = strlen(e);
        if (s.size() > n && s.compare(s.size() - n, n, e) == 0) s.resize(s.size() - n);
    }
    return s;
}

// union of the -m tracks that exist for this DB (a missing track is reported and skipped, the
// way daligner treats a track that was never computed for a block)
static void apply_masks(const std::string &arg, OpenDb &o, const std::vector<std::string> &tracks)
{
    const int32_t n = dh_dazz_nreads(o.dz);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> per((size_t)n);
    bool any = false;
    for (const std::string &t : tracks) {
        std::vector<int64_t> ptr((size_t)n + 1);
        const int64_t m = dh_dazz_read_mask(o.dz, arg.c_str(), t.c_str(), ptr.data(), nullptr, 0);
        if (m < 0) {
            fprintf(stderr, "mask track `%s` not found for %s: skipped\n", t.c_str(), arg.c_str());
            continue;
        }
        std::vector<int32_t> iv((size_t)std::max<int64_t>(2 * m, 2));
        dh_dazz_read_mask(o.dz, arg.c_str(), t.c_str(), ptr.data(), iv.data(), m);
        for (int32_t i = 0; i < n; i++)
            for (int64_t j = ptr[(size_t)i]; j < ptr[(size_t)i + 1]; j++) per[(size_t)i].push_back({iv[(size_t)(2 * j)], iv[(size_t)(2 * j + 1)]});
        any = true;
    }
    if (!any) return;
    std::vector<int64_t> ptr((size_t)n + 1, 0);
    std::vector<int32_t> iv;
    for (int32_t i = 0; i < n; i++) {
        auto &v = per[(size_t)i];
        std::sort(v.begin(), v.end());
        int32_t cb = -1, ce = -1;
        for (auto &x : v) {
            if (cb >= 0 && x.first <= ce) {
                ce = std::max(ce, x.second);
                continue;
            }
            if (cb >= 0) {
                iv.push_back(cb);
                iv.push_back(ce);
            }
            cb = x.first;
            ce = x.second;
        }
        if (cb >= 0) {
            iv.push_back(cb);
            iv.push_back(ce);
        }
        ptr[(size_t)i + 1] = (int64_t)iv.size() / 2;
    }
    iv.push_back(0);
    CHK(dh_db_set_mask(o.dev, ptr.data(), iv.data()));
}

static OpenDb open_db(dh_ctx *ctx, const std::string &arg, const std::vector<std::string> &tracks)
{
    OpenDb o;
    o.name = las_name_part(arg);
    CHK(dh_dazz_open(arg.c_str(), &o.dz));
    CHK(dh_db_create(ctx, dh_dazz_bases(o.dz), dh_dazz_offsets(o.dz), dh_dazz_nreads(o.dz), nullptr, &o.dev));
    if (!tracks.empty()) apply_masks(arg, o, tracks);
    return o;
}

// write one .las: ids are trimmed DB ids (block first id + local index)
static void write_las(const std::string &path, dh_la_set *set, int afirst, int bfirst, int tspace)
{
    const int64_t n = dh_la_set_count(set);
    // chains that damapper's -n rule discards come back DISABLED: they are not written
    std::vector<dh_la> las;
    las.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        dh_la l = dh_la_set_records(set)[i];
        if (l.flags & DH_FLAG_DISABLED) continue;
        l.aread += afirst;
        l.bread += bfirst;
        las.push_back(l);
    }
    CHK(dh_las_write(path.c_str(), las.data(), (int64_t)las.size(), dh_la_set_trace(set), tspace));
}

int main(int argc, char **argv)
{
    std::string mode = argv[0];
    const size_t slash = mode.find_last_of('/');
    if (slash != std::string::npos) mode = mode.substr(slash + 1);
    dh_align_opts o;
    dh_default_align_opts(&o);
    bool flagA = false, flagI = false, flagC = false, verbose = false, seen_k = false, seen_w = false;
    double e = 0.7, near_best = 1.0;  // damapper's own -n default is 1.00 (best chains only); DENTIST always passes -n.7
    std::vector<std::string> dbs, tracks;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--mode" && i + 1 < argc) {
            mode = argv[++i];
            continue;
        }
        if (a.size() < 2 || a[0] != '-') {
            dbs.push_back(a);
            continue;
        }
        const char *v = a.c_str() + 2;
        switch (a[1]) {
        case 'k': o.k = atoi(v); seen_k = true; break;
        case 'w': o.band_shift = atoi(v); seen_w = true; break;
        case 'h': o.hmin = atoi(v); break;
        case 't': o.tcap = atoi(v); break;
        case 's': o.tspace = atoi(v); break;
        case 'l': o.min_len = atoi(v); break;
        case 'e': e = atof(v); break;
        case '%': o.kmer_mod = atoi(v); break;
        case 'A': flagA = true; break;
        case 'I': flagI = true; break;
        case 'C': flagC = true; break;
        case 'v': verbose = true; break;
        case 'm': tracks.push_back(v); break;
        case 'T': case 'P': case 'M': break;  // threads / temp dir / memory: no meaning on the device
        case 'n': near_best = atof(v); break;
        case 'B': case 'b': case 'p': case 'z': case 'H':
            // accepted for DENTIST's argv, but not implemented: say so instead of silently differing
            fprintf(stderr, "%s: option %s is accepted but has no effect in this implementation\n", mode.c_str(), a.c_str());
            break;
        default: die(mode + ": unknown option " + a);
        }
    }
    if (mode.find("datander") != std::string::npos) {
        if (dbs.empty()) die("usage: datander [-v] [-k<int(12)>] [-w<int(4)>] [-h<int(35)>] [-T<int(4)>] [-P<dir>] [-e<double(.70)>] [-l<int(500)>] [-s<int(100)>] <path:db|dam> ...");
    } else if (dbs.size() < 2)
        die("usage: " + mode + " [-k -w -h -t -s -l -e -A -I -C -T -m...] <subject:db|dam> <target:db|dam> ...");
    if (e <= 0.5 || e >= 1.0) die(mode + ": -e must be in (0.5, 1)");
    o.pen = (int)floor(2.0 / (1.0 - e));
    o.max_err_ppm = (int)llround((1.0 - e) * 1e6);
    const bool mapper = mode.find("damapper") != std::string::npos;
    const bool tander = mode.find("datander") != std::string::npos;
    if (mapper && o.min_len == 500 && o.tspace == 100) o.min_len = 500;
    if (mapper) {
        // damapper: the tiled band extension (one alignment per lane) and chains with the -n near-best rule
        o.algo = 1;
        o.width = 64;
        if (near_best < 0 || near_best > 1) die(mode + ": -n must be in [0, 1]");
    }

    dh_ctx *ctx = nullptr;
    CHK(dh_ctx_create(0, nullptr, &ctx));
    if (tander) {
        // datander <block> ...: every read of a block against ITSELF, the local alignments off the main diagonal (tandem
        // repeats) -> TAN.<block>.las (DAMASKER; DENTIST's call: `datander -T<n> -s126 -l500 -e0.7 <dam>.<block>`,
        // commandline.d:2866-2876, snakemake/Snakefile:1056-1076; TANmask reads the file).  Its own defaults: -k12 -w4 -h35.
        if (!seen_k) o.k = 12;
        if (!seen_w) o.band_shift = 4;
        o.algo = 1;
        o.width = 64;
        o.strands = 1;
        o.skip_self = 3;
        for (const std::string &arg : dbs) {
            OpenDb A = open_db(ctx, arg, tracks);
            dh_la_set *set = nullptr;
            CHK(dh_align_db(ctx, A.dev, A.dev, &o, 0, &set));
            write_las("TAN." + A.name + ".las", set, dh_dazz_first_id(A.dz), dh_dazz_first_id(A.dz), o.tspace);
            if (verbose) fprintf(stderr, "datander: %lld local alignments -> TAN.%s.las\n", (long long)dh_la_set_count(set), A.name.c_str());
            dh_la_set_destroy(set);
            dh_db_destroy(A.dev);
            dh_dazz_close(A.dz);
        }
        dh_ctx_destroy(ctx);
        return 0;
    }
    if (mapper) CHK(dh_ctx_set_near_best(ctx, (int32_t)llround(near_best * 1e6)));
    OpenDb A = open_db(ctx, dbs[0], tracks);
    for (size_t bi = 1; bi < dbs.size(); bi++) {
        const bool same = las_name_part(dbs[bi]) == A.name;
        OpenDb B = same ? A : open_db(ctx, dbs[bi], tracks);
        dh_align_opts oo = o;
        // one DB against itself: every unordered pair once, both records written (daligner's own
        // behaviour); -I also aligns a read against itself, which needs the plain all-vs-all
        oo.skip_self = (same && !flagI) ? 2 : 0;
        if (oo.skip_self == 2) {
            oo.max_la = std::max(oo.max_la, 64);
            oo.max_cand = std::max(oo.max_cand, 128);
        }
        dh_la_set *ab = nullptr, *ba = nullptr;
        // damapper -C, and daligner on two DBs without -A (which writes A.B.las and B.A.las, dazzler.d:6121-6140; the
        // workflow's block pairs `daligner -I ... ref.i ref.j`, Snakefile:998-1022): the second file comes out of the same
        // pass -- the transposed pair of every alignment, through its seed (dh_align_db_transposed; DH-2, the tiled band)
        const bool both = !same && !mapper && !flagA;
        if (both) {
            oo.algo = 1;
            oo.width = 64;
        }
        const bool transposed = !same && ((mapper && flagC) || both);
        if (transposed)
            CHK(dh_align_db_transposed(ctx, A.dev, B.dev, &oo, mapper ? 1 : 0, &ab, &ba));
        else
            CHK(dh_align_db(ctx, A.dev, B.dev, &oo, mapper ? 1 : 0, &ab));
        write_las(A.name + "." + B.name + ".las", ab, dh_dazz_first_id(A.dz), dh_dazz_first_id(B.dz), o.tspace);
        if (verbose) fprintf(stderr, "%s: %lld local alignments -> %s.%s.las\n", mode.c_str(), (long long)dh_la_set_count(ab), A.name.c_str(), B.name.c_str());
        dh_la_set_destroy(ab);
        if (ba) {
            write_las(B.name + "." + A.name + ".las", ba, dh_dazz_first_id(B.dz), dh_dazz_first_id(A.dz), o.tspace);
            dh_la_set_destroy(ba);
        }
        if (!same) {
            dh_db_destroy(B.dev);
            dh_dazz_close(B.dz);
        }
    }
    dh_db_destroy(A.dev);
    dh_dazz_close(A.dz);
    dh_ctx_destroy(ctx);
    return 0;
}
