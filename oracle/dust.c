/*
 * dust.c -- oracle restatement of the low-complexity mask (the role of DBdust: symmetric DUST with
 * the upstream defaults -w64 -t2.0 -m10; DENTIST runs it on every DB it aligns with -mdust,
 * source/dentist/commands/processPileUps/package.d:476-482, 655-667, flags commandline.d:2904-2907).
 *
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h).  PARITY UNPINNED: DBdust's source (DAZZ_DB d22ae58) is
 * not under /root/reference and the reference holds no dust vectors.  Rule restated from the DUST
 * score (Morgulis et al. 2006): a window of L = 16, 32 or 64 bases is low-complexity when
 *     S = sum over triplet codes t of c_t (c_t - 1) / 2  >  2 (l - 1),   l = L - 2 triplets
 * (score above 2.0); windows holding a non-ACGT base are skipped; the mask is the union of all such
 * windows.  Reverse-complement symmetric: S depends on the multiset of triplets only.
 */
#include "dh_oracle.h"

#include <stdlib.h>
#include <string.h>

static int trip(const uint8_t *b, int64_t i)
{
    return (b[i] | b[i + 1] | b[i + 2]) > 3 ? -1 : (b[i] << 4 | b[i + 1] << 2 | b[i + 2]);
}

/* flags[len]: 1 = masked */
static void dust_flags(const uint8_t *b, int64_t len, uint8_t *flags)
{
    static const int Ls[3] = {16, 32, 64};
    memset(flags, 0, (size_t)len);
    for (int li = 0; li < 3; li++) {
        const int L = Ls[li];
        if (len < L) continue;
        int cnt[64] = {0};
        int S = 0, bad = 0;
        for (int64_t i = 0; i < L - 2; i++) {
            const int c = trip(b, i);
            if (c < 0)
                bad++;
            else
                S += cnt[c]++;
        }
        for (int64_t a = 0; a + L <= len; a++) {
            if (bad == 0 && S > 2 * (L - 3)) memset(flags + a, 1, (size_t)L);
            if (a + L < len) {
                const int c0 = trip(b, a), c1 = trip(b, a + L - 2);
                if (c0 < 0)
                    bad--;
                else
                    S -= --cnt[c0];
                if (c1 < 0)
                    bad++;
                else
                    S += cnt[c1]++;
            }
        }
    }
}

/* mask of every sequence of db as intervals: ptr[n + 1] (caller), *iv_out malloc'd (begin, end)
 * pairs (free with oz_free); returns the number of intervals */
int64_t oz_dust(const oz_db *db, int64_t *ptr, int32_t **iv_out)
{
    int64_t cap = 1024, m = 0;
    int32_t *iv = (int32_t *)malloc((size_t)cap * 2 * sizeof(int32_t));
    int64_t maxlen = 1;
    for (int32_t s = 0; s < db->n; s++)
        if (db->off[s + 1] - db->off[s] > maxlen) maxlen = db->off[s + 1] - db->off[s];
    uint8_t *flags = (uint8_t *)malloc((size_t)maxlen);
    for (int32_t s = 0; s < db->n; s++) {
        ptr[s] = m;
        const int64_t len = db->off[s + 1] - db->off[s];
        dust_flags(db->bases + db->off[s], len, flags);
        for (int64_t g = 0; g < len;) {
            if (!flags[g]) {
                g++;
                continue;
            }
            int64_t h = g;
            while (h < len && flags[h]) h++;
            if (m == cap) {
                cap *= 2;
                iv = (int32_t *)realloc(iv, (size_t)cap * 2 * sizeof(int32_t));
            }
            iv[2 * m] = (int32_t)g;
            iv[2 * m + 1] = (int32_t)h;
            m++;
            g = h;
        }
    }
    ptr[db->n] = m;
    free(flags);
    *iv_out = iv;
    return m;
}
