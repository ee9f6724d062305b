/*
 * nw.c -- oracle: Needleman-Wunsch with unit mismatch score and constant indel penalty.
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h).
 *
 * Follows source/dentist/util/string.d:478-520 (`findAlignment`: matrix fill, optional
 * freeShift initialisation of row 0 / column 0 with zeros) and :775-831
 * (`tracebackScoringMatrix`: from the bottom-right corner step to the neighbour with the
 * smallest score, preferring diagonal, then insertion (j-1), then deletion (i-1); leftover
 * rows become deletions, leftover columns insertions).  PINNED by the golden alignments of
 * util/string.d:523-751 (tests/golden/nw_cases.json).
 */
#include "dh_oracle.h"

#include <stdlib.h>

uint32_t oz_nw(const uint8_t *ref, int32_t rlen, const uint8_t *qry, int32_t qlen,
               uint32_t indel, int free_shift, uint8_t *ops, int32_t *nops)
{
    const size_t W = (size_t)qlen + 1;
    uint32_t *F = (uint32_t *)malloc(((size_t)rlen + 1) * W * sizeof(uint32_t));
    for (int32_t i = 0; i <= rlen; i++) F[(size_t)i * W] = free_shift ? 0 : (uint32_t)i * indel;
    for (int32_t j = 0; j <= qlen; j++) F[j] = free_shift ? 0 : (uint32_t)j * indel;
    for (int32_t i = 1; i <= rlen; i++)
        for (int32_t j = 1; j <= qlen; j++) {
            uint32_t m = F[(size_t)(i - 1) * W + (j - 1)] + (ref[i - 1] == qry[j - 1] ? 0u : 1u);
            uint32_t d = F[(size_t)(i - 1) * W + j] + indel;
            uint32_t s = F[(size_t)i * W + (j - 1)] + indel;
            uint32_t v = m < d ? m : d;
            F[(size_t)i * W + j] = v < s ? v : s;
        }
    const uint32_t score = F[(size_t)rlen * W + qlen];

    const int32_t cap = rlen + qlen;
    uint8_t *tmp = (uint8_t *)malloc((size_t)(cap ? cap : 1));
    int32_t k = cap, i = rlen, j = qlen;
    while (i > 0 && j > 0) {
        const uint32_t ms = F[(size_t)(i - 1) * W + (j - 1)];
        const uint32_t is = F[(size_t)i * W + (j - 1)];
        const uint32_t ds = F[(size_t)(i - 1) * W + j];
        uint32_t nx = ms < ds ? ms : ds;
        if (is < nx) nx = is;
        if (nx == ms) {
            tmp[--k] = OZ_OP_SUB;
            --i;
            --j;
        } else if (nx == is) {
            tmp[--k] = OZ_OP_INS;
            --j;
        } else {
            tmp[--k] = OZ_OP_DEL;
            --i;
        }
    }
    while (i > 0) {
        tmp[--k] = OZ_OP_DEL;
        --i;
    }
    while (j > 0) {
        tmp[--k] = OZ_OP_INS;
        --j;
    }
    *nops = cap - k;
    for (int32_t t = 0; t < *nops; t++) ops[t] = tmp[k + t];
    free(tmp);
    free(F);
    return score;
}
