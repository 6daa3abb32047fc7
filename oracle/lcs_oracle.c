/*
 * lcs_oracle.c -- CPU restatement of FAMSA's bit-parallel LCS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (famsa_amd/, the C-ABI
 * library, the host tools) may call, link or load this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (1) the reference's own golden files test/adeno_fiber/{pid,pid_sq,dist,
 *       dist_sq}.csv (copied as data under tests/golden/adeno_fiber/), and
 *   (2) LCS matrices produced by the reference's own CLCSBP (classic and AVX2
 *       dispatch) compiled from /root/reference by oracle/Makefile into
 *       oracle/_ref/ (fixtures under tests/golden/, generator script
 *       oracle/make_golden.py).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src).  The code is written from the recurrence, not from the
 * reference text.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_UNKNOWN_SYMBOL 22u /* core/defs.h:66 */
#define ORACLE_N_VALID 20u        /* core/defs.h:71 NO_VALID_AMINOACIDS */
#define ORACLE_N_SYMBOLS 32u      /* core/defs.h:68 NO_SYMBOLS */

/* Residue character -> symbol code.
 * Restates CSequence::CSequence, core/sequence.cpp:17 (alphabet) and :53-79:
 * '-' is dropped by the caller, chars above 'Z' are shifted down by 32, the
 * code is the index in "ARNDCQEGHILKMFPSTWYVBZX*" (the search range is 25 wide,
 * so a NUL byte maps to 24), anything else is UNKNOWN_SYMBOL (22). */
int oracle_encode_char(int ch)
{
    static const char table[25] = "ARNDCQEGHILKMFPSTWYVBZX*";
    char c = (char)ch;
    if (c > 'Z')
        c = (char)(c - 32);
    for (int i = 0; i < 25; ++i)
        if (table[i] == c)
            return i;
    return (int)ORACLE_UNKNOWN_SYMBOL;
}

/* Encode a residue string (gaps '-' removed). Returns the number of codes.
 * core/sequence.cpp:30-79. */
size_t oracle_encode(const char *residues, size_t n, uint8_t *codes)
{
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (residues[i] == '-')
            continue;
        codes[m++] = (uint8_t)oracle_encode_char(residues[i]);
    }
    return m;
}

/* Per-symbol occurrence masks of the ref, M[c*bv_len + w] bit (i%64) set iff
 * ref[i]==c and c<20.  Restates CSequence::ComputeBitMasks,
 * core/sequence.cpp:190-201 (only codes 0..19 get bits; the table has 32 rows).
 * The stride used here is bv_len = ceil(len/64); the reference strides by the
 * padded data_size but only the first ceil(length/64) words are ever read
 * (lcs/lcsbp_classic.cpp:42). */
static void build_masks(const uint8_t *ref, uint32_t len, uint32_t bv_len, uint64_t *M)
{
    memset(M, 0, (size_t)ORACLE_N_SYMBOLS * bv_len * sizeof(uint64_t));
    for (uint32_t i = 0; i < len; ++i)
        if (ref[i] < ORACLE_N_VALID)
            M[(size_t)ref[i] * bv_len + i / 64] |= 1ull << (i % 64);
}

/* The LCS-like score of FAMSA: ref is the bit-mask side, partner is streamed.
 * Restates CLCSBP_Classic::Calculate + CLCSBP_Classic_Impl::LoopCalculate,
 * lcs/lcsbp_classic.cpp:39-86 and lcs/lcsbp_classic.h:51-98:
 *   X[w] = ~0; for each partner residue c (skip c==22): cin = 0;
 *     for w: V = X[w]; tB = V & M[c][w]; V2 = V + tB + cin;
 *            cin = (V2 < V);            <- NOT a true carry (SURVEY note Q)
 *            X[w] = V2 | (V - tB);
 *   result = sum_w popcount(~X[w]).
 * Partner codes outside 0..31 cannot occur (encode yields 0..24). */
uint32_t oracle_lcs_with_masks(const uint64_t *M, uint32_t bv_len, const uint8_t *partner,
                               uint32_t len_p, uint64_t *X)
{
    for (uint32_t w = 0; w < bv_len; ++w)
        X[w] = ~0ull;
    for (uint32_t i = 0; i < len_p; ++i) {
        uint32_t c = partner[i];
        if (c == ORACLE_UNKNOWN_SYMBOL)
            continue;
        const uint64_t *Mc = M + (size_t)c * bv_len;
        uint64_t cin = 0;
        for (uint32_t w = 0; w < bv_len; ++w) {
            uint64_t V = X[w];
            uint64_t tB = V & Mc[w];
            uint64_t V2 = V + tB + cin;
            cin = (V2 < V) ? 1u : 0u;
            X[w] = V2 | (V - tB);
        }
    }
    uint32_t res = 0;
    for (uint32_t w = 0; w < bv_len; ++w)
        res += (uint32_t)__builtin_popcountll(~X[w]);
    return res;
}

uint32_t oracle_lcs(const uint8_t *ref, uint32_t len_ref, const uint8_t *partner, uint32_t len_p)
{
    uint32_t bv_len = (len_ref + 63) / 64;
    if (bv_len == 0)
        return 0;
    uint64_t *M = (uint64_t *)malloc((size_t)ORACLE_N_SYMBOLS * bv_len * sizeof(uint64_t));
    uint64_t *X = (uint64_t *)malloc((size_t)bv_len * sizeof(uint64_t));
    build_masks(ref, len_ref, bv_len, M);
    uint32_t r = oracle_lcs_with_masks(M, bv_len, partner, len_p, X);
    free(M);
    free(X);
    return r;
}

/* Rectangle of oriented LCS values: out[r*n_cols + c] = LCS(ref = seq ref_ids[r],
 * partner = seq col_ids[c]).  The batch shape of calculateDistanceVector /
 * calculateDistanceRange (tree/AbstractTreeGenerator.hpp:130-279) without the
 * Transform.  codes/offsets: concatenated symbol codes, offsets[n+1]. */
void oracle_lcs_rect(const uint8_t *codes, const uint64_t *offsets, const int32_t *ref_ids,
                     int32_t n_refs, const int32_t *col_ids, int32_t n_cols, uint32_t *out)
{
    for (int32_t r = 0; r < n_refs; ++r) {
        const uint8_t *ref = codes + offsets[ref_ids[r]];
        uint32_t len_ref = (uint32_t)(offsets[ref_ids[r] + 1] - offsets[ref_ids[r]]);
        uint32_t bv_len = (len_ref + 63) / 64;
        if (bv_len == 0) {
            for (int32_t c = 0; c < n_cols; ++c)
                out[(size_t)r * n_cols + c] = 0;
            continue;
        }
        uint64_t *M = (uint64_t *)malloc((size_t)ORACLE_N_SYMBOLS * bv_len * sizeof(uint64_t));
        uint64_t *X = (uint64_t *)malloc((size_t)bv_len * sizeof(uint64_t));
        build_masks(ref, len_ref, bv_len, M);
        for (int32_t c = 0; c < n_cols; ++c) {
            const uint8_t *p = codes + offsets[col_ids[c]];
            uint32_t len_p = (uint32_t)(offsets[col_ids[c] + 1] - offsets[col_ids[c]]);
            out[(size_t)r * n_cols + c] = oracle_lcs_with_masks(M, bv_len, p, len_p, X);
        }
        free(M);
        free(X);
    }
}

/* Lower triangle, row-major, index i*(i-1)/2 + j for j < i, ref = row i,
 * partner = column j.  calculateDistanceMatrix, AbstractTreeGenerator.hpp:378-398
 * with TriangleMatrix::access, tree/TreeDefs.h:115-120. */
void oracle_lcs_triangle(const uint8_t *codes, const uint64_t *offsets, int32_t n, uint32_t *out)
{
    for (int32_t i = 1; i < n; ++i) {
        const uint8_t *ref = codes + offsets[i];
        uint32_t len_ref = (uint32_t)(offsets[i + 1] - offsets[i]);
        uint32_t bv_len = (len_ref + 63) / 64;
        size_t row = (size_t)i * (size_t)(i - 1) / 2;
        if (bv_len == 0) {
            for (int32_t j = 0; j < i; ++j)
                out[row + j] = 0;
            continue;
        }
        uint64_t *M = (uint64_t *)malloc((size_t)ORACLE_N_SYMBOLS * bv_len * sizeof(uint64_t));
        uint64_t *X = (uint64_t *)malloc((size_t)bv_len * sizeof(uint64_t));
        build_masks(ref, len_ref, bv_len, M);
        for (int32_t j = 0; j < i; ++j) {
            const uint8_t *p = codes + offsets[j];
            uint32_t len_p = (uint32_t)(offsets[j + 1] - offsets[j]);
            out[row + j] = oracle_lcs_with_masks(M, bv_len, p, len_p, X);
        }
        free(M);
        free(X);
    }
}

/* A plain O(nm) dynamic-programming LCS over the same match relation (a==b and
 * a<20).  Not part of the reference: an independent cross-check that the
 * bit-parallel restatement equals the true LCS on inputs that cannot trigger
 * the carry quirk. */
uint32_t oracle_lcs_dp(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb)
{
    uint32_t *row = (uint32_t *)calloc((size_t)lb + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < la; ++i) {
        uint32_t diag = 0;
        for (uint32_t j = 0; j < lb; ++j) {
            uint32_t up = row[j + 1];
            uint32_t v;
            if (a[i] == b[j] && a[i] < ORACLE_N_VALID)
                v = diag + 1;
            else
                v = up > row[j] ? up : row[j];
            diag = up;
            row[j + 1] = v;
        }
    }
    uint32_t r = row[lb];
    free(row);
    return r;
}

/* ---- LCS -> distance, tree/AbstractTreeGenerator.hpp:37-82 ---------------- */

/* Transform<double, indel075_div_lcs>, hpp:37-63: pow(indel, 0.75) computed in
 * double from the uint32 indel (table entry (T)pow(i, 0.75)), divided by
 * (double)lcs; lcs==0 -> nextafter(DBL_MAX, 0). */
double oracle_dist_indel075_f64(uint32_t lcs, uint32_t len1, uint32_t len2)
{
    uint32_t indel = len1 + len2 - 2 * lcs;
    if (lcs == 0)
        return nextafter(DBL_MAX, 0.0);
    return pow((double)indel, 0.75) / (double)lcs;
}

/* Transform<float, indel075_div_lcs>: table entry is (float)pow((double)i,0.75),
 * the division is float/float.  Note hpp:53 computes indel as (float)(uint32)
 * and indexes the table with it, exact for indel < 2^24. */
float oracle_dist_indel075_f32(uint32_t lcs, uint32_t len1, uint32_t len2)
{
    uint32_t indel = len1 + len2 - 2 * lcs;
    if (lcs == 0)
        return (float)nextafter((double)FLT_MAX, 0.0); /* hpp:61: the int 0 promotes both operands to double */
    float p = (float)pow((double)indel, 0.75);
    return p / (float)lcs;
}

/* Transform<double, indel_div_lcs>, hpp:65-75: (double)indel / lcs. */
double oracle_dist_indel_f64(uint32_t lcs, uint32_t len1, uint32_t len2)
{
    uint32_t indel = len1 + len2 - 2 * lcs;
    if (lcs == 0)
        return nextafter(DBL_MAX, 0.0);
    return (double)indel / (double)lcs;
}

float oracle_dist_indel_f32(uint32_t lcs, uint32_t len1, uint32_t len2)
{
    uint32_t indel = len1 + len2 - 2 * lcs;
    if (lcs == 0)
        return (float)nextafter((double)FLT_MAX, 0.0);
    return (float)indel / (float)lcs;
}

/* calculateDistanceMatrix<.., float, ..> (hpp:378-398) from an LCS triangle: out[i(i-1)/2 + j],
 * ref = i, partner = j < i.  kind 0 = indel_div_lcs, 1 = indel075_div_lcs. */
void oracle_dist_triangle_f32(const uint32_t *lcs, const uint32_t *lens, int32_t n, int kind, float *out)
{
    size_t k = 0;
    for (int32_t i = 1; i < n; ++i)
        for (int32_t j = 0; j < i; ++j, ++k)
            out[k] = kind == 1 ? oracle_dist_indel075_f32(lcs[k], lens[i], lens[j])
                               : oracle_dist_indel_f32(lcs[k], lens[i], lens[j]);
}

/* Transform<float, pairwise_identity>, hpp:77-82: (float)lcs / min(len1,len2). */
float oracle_pid_f32(uint32_t lcs, uint32_t len1, uint32_t len2)
{
    uint32_t m = len1 < len2 ? len1 : len2;
    return (float)lcs / (float)m;
}

/* NumericConversions::Double2PChar(val, 6), utils/conversion.h:109-119, as used
 * by the -dist_export writer (tree/DistanceCalculator.cpp:102-105): integer part,
 * then (1+frac)*1e6+0.5 printed with its leading '1' replaced by '.'.
 * Returns the number of characters written (no terminator). */
/* double -> int64 with the x86-64 cvttsd2si result the reference binary gets
 * for out-of-range values (the "integer indefinite" 0x8000000000000000); in
 * range it is the plain C truncation. Needed for the lcs==0 distance. */
static int64_t trunc_i64_x86(double v)
{
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0))
        return INT64_MIN;
    return (int64_t)v;
}

static int put_u64(uint64_t v, char *str)
{
    char tmp[32];
    int n = 0, k = 0;
    if (v == 0)
        tmp[n++] = '0';
    for (; v > 0; v /= 10)
        tmp[n++] = (char)('0' + v % 10);
    while (n > 0)
        str[k++] = tmp[--n];
    return k;
}

int oracle_format_dist(double val, char *str)
{
    int64_t a = trunc_i64_x86(val);
    int64_t b = trunc_i64_x86((1.0 + (val - (double)a)) * 1000000.0 + 0.5);
    int r1 = put_u64((uint64_t)a, str);
    int r2 = put_u64((uint64_t)b, str + r1);
    str[r1] = '.';
    return r1 + r2;
}
