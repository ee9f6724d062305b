/*
 * seq_las_trace.c -- oracle: base codes, .las codec, trace-point translation.
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h).
 */
#include "dh_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* DAZZ_DB code order a,c,g,t = 0..3 (reverse complement semantics of
 * source/dentist/util/fasta.d:563-604: complement and reverse, case-insensitive). */
void oz_encode(const char *ascii, int64_t n, uint8_t *codes)
{
    for (int64_t i = 0; i < n; i++) {
        switch (ascii[i]) {
        case 'a': case 'A': codes[i] = 0; break;
        case 'c': case 'C': codes[i] = 1; break;
        case 'g': case 'G': codes[i] = 2; break;
        case 't': case 'T': codes[i] = 3; break;
        default: codes[i] = 4;
        }
    }
}

void oz_decode(const uint8_t *codes, int64_t n, char *ascii)
{
    static const char lut[5] = {'a', 'c', 'g', 't', 'n'};
    for (int64_t i = 0; i < n; i++) ascii[i] = lut[codes[i] > 4 ? 4 : codes[i]];
}

void oz_revcomp(const uint8_t *src, int64_t n, uint8_t *dst)
{
    for (int64_t i = 0; i < n; i++) {
        uint8_t c = src[n - 1 - i];
        dst[i] = c < 4 ? (uint8_t)(3 - c) : c;
    }
}

/* ------------------------------------------------------------------ .las ---------------
 * Header: int64 novl, int32 tspace (dazzler.d:1665-1689).  Record: bytes [8,48) of
 * struct { void* trace; int tlen,diffs,abpos,bbpos,aepos,bepos; uint flags; int aread,bread; }
 * i.e. 9 x int32 + 4 pad bytes = 40 (dazzler.d:1716-1725, 1988-2016); then tlen trace values,
 * u8 when tspace <= 125 (TRACE_XOVR) else u16 (dazzler.d:2019-2025, 2130-2170). */

int oz_las_write(const char *path, const oz_la_set *s, int32_t tspace)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int64_t novl = s->n;
    fwrite(&novl, 8, 1, f);
    fwrite(&tspace, 4, 1, f);
    const int small = tspace <= 125;
    for (int64_t i = 0; i < s->n; i++) {
        const oz_la *la = &s->la[i];
        int32_t rec[10] = {la->tlen, la->diffs, la->abpos, la->bbpos, la->aepos,
                           la->bepos, (int32_t)la->flags, la->aread, la->bread, 0};
        fwrite(rec, 4, 10, f);
        const uint16_t *t = s->trace + la->toff;
        if (small) {
            for (int32_t j = 0; j < la->tlen; j++) {
                uint8_t v = (uint8_t)t[j];
                fwrite(&v, 1, 1, f);
            }
        } else
            fwrite(t, 2, (size_t)la->tlen, f);
    }
    fclose(f);
    return 0;
}

int oz_las_read(const char *path, oz_la_set *s, int32_t *tspace)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int64_t novl;
    int32_t ts;
    if (fread(&novl, 8, 1, f) != 1 || fread(&ts, 4, 1, f) != 1) {
        fclose(f);
        return -2;
    }
    *tspace = ts;
    const int small = ts <= 125;
    oz_la_set_init(s);
    s->cap = novl ? novl : 1;
    s->la = (oz_la *)calloc((size_t)s->cap, sizeof(oz_la));
    s->tcap = 4096;
    s->trace = (uint16_t *)malloc((size_t)s->tcap * 2);
    for (int64_t i = 0; i < novl; i++) {
        int32_t rec[10];
        if (fread(rec, 4, 10, f) != 10) {
            fclose(f);
            return -3; /* unexpected EOF: dazzler.d:1825-1833 */
        }
        oz_la *la = &s->la[i];
        la->tlen = rec[0];
        la->diffs = rec[1];
        la->abpos = rec[2];
        la->bbpos = rec[3];
        la->aepos = rec[4];
        la->bepos = rec[5];
        la->flags = (uint32_t)rec[6];
        la->aread = rec[7];
        la->bread = rec[8];
        if (la->tlen % 2) {
            fclose(f);
            return -4; /* "illegal value for tlen" dazzler.d:1769-1772 */
        }
        la->toff = s->tn;
        while (s->tn + la->tlen > s->tcap) {
            s->tcap *= 2;
            s->trace = (uint16_t *)realloc(s->trace, (size_t)s->tcap * 2);
        }
        if (small) {
            for (int32_t j = 0; j < la->tlen; j++) {
                uint8_t v;
                if (fread(&v, 1, 1, f) != 1) {
                    fclose(f);
                    return -3;
                }
                s->trace[s->tn + j] = v;
            }
        } else if (fread(s->trace + s->tn, 2, (size_t)la->tlen, f) != (size_t)la->tlen) {
            fclose(f);
            return -3;
        }
        s->tn += la->tlen;
        s->n++;
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ trace --------------
 * source/dentist/common/alignments/base.d:169-299 */

static int32_t floor_to(int32_t x, int32_t m) { return x / m * m; }
static int32_t ceildiv(int32_t a, int32_t b) { return (a + b - 1) / b; }

/* tracePointsUpTo!"contigA"  base.d:203-237 */
int32_t oz_trace_points_up_to_a(int32_t abpos, int32_t aepos, int32_t tspace, int32_t ntp,
                                int32_t apos, int mode)
{
    const int32_t first = abpos;
    const int32_t second = floor_to(first, tspace) + tspace;
    const int32_t second_from_last = floor_to(aepos - 1, tspace);
    if (mode == OZ_FLOOR) {
        if (apos < second) return 0;
        if (apos < aepos) return 1 + (apos - second) / tspace;
        return ntp;
    }
    if (first == apos) return 0;
    if (apos <= second) return 1;
    if (apos <= second_from_last) return 1 + ceildiv(apos - second, tspace);
    return ntp;
}

/* tracePointsUpTo!"contigB"  base.d:265-298 */
int32_t oz_trace_points_up_to_b(int32_t bbpos, int32_t bepos, const uint16_t *trace, int32_t ntp,
                                int32_t bpos, int mode)
{
    if (bpos == bbpos) return 0;
    if (bpos == bepos) return ntp;
    int32_t acc = bbpos;
    /* positions: index 0 -> bbpos, index i -> bbpos + sum(bbases[0..i)) */
    for (int32_t idx = 0; idx <= ntp; idx++) {
        if (idx > 0) acc += trace[2 * (idx - 1) + 1];
        if (mode == OZ_FLOOR) {
            if (bpos < acc) return idx - 1;
        } else if (bpos <= acc)
            return idx;
    }
    return ntp;
}

/* translateTracePoint!"contigA"  base.d:185-201 */
void oz_translate_trace_point_a(int32_t abpos, int32_t aepos, int32_t bbpos, int32_t tspace,
                                const uint16_t *trace, int32_t ntp, int32_t apos, int mode,
                                int32_t *outa, int32_t *outb)
{
    const int32_t idx = oz_trace_points_up_to_a(abpos, aepos, tspace, ntp, apos, mode);
    int32_t b = bbpos;
    for (int32_t i = 0; i < idx; i++) b += trace[2 * i + 1];
    *outb = b;
    *outa = idx == 0 ? abpos : (idx < ntp ? floor_to(abpos, tspace) + idx * tspace : aepos);
}
