/*
 * align_oracle.c — CPU restatement of the reference's alignment step (`-aln`), TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under oracle/; the product
 * (kaamer_amd/csrc/align.hip) never does.
 *
 * What is restated, and what pins it:
 *
 *   PINNED by the reference's own files (every line cited):
 *     pkg/align/align.go:46-161     Align(): [uU] -> '*', the fixed call SWAffine{Matrix: BLOSUM62, GapOpen: -11}
 *                                   (whatever the options say), the identity / similarity / mismatch loop in float32,
 *                                   AlnString, Length, the raw score and gap openings from the feature pairs
 *                                   (a feature whose Score() == -GapOpen is a gap; (len - 1) * GapExtend more),
 *                                   BitScore = (lambda * raw - ln K) / ln 2, EValue = len(query) * NumberOfAA / 2^BitScore,
 *                                   Query/Subject Start (+1) and End
 *     pkg/align/matrixScores.go:22-127  the (matrix, open, extend) -> (lambda, K) table, the key format, "No matrix
 *                                   found", AAPosInMatrix, GetAlnScoreAA
 *     pkg/search/search.go:483-494  the call site: one Align per reported hit, hits re-sorted by BitScore
 *
 *   NOT in /root/reference — third-party dependency, restated from its published algorithm: PARITY UNPINNED
 *     github.com/biogo/biogo v1.0.1 (go.mod:8): align.SWAffine, align.Format, matrix.BLOSUM62, alphabet.Protein.
 *       - the recurrence: three layers (diag / up / left) of a local alignment, an affine gap = GapOpen on entering a
 *         gap layer + the matrix's gap-column score per gap position; scores floored at zero;
 *       - the alphabet "-ABCDEFGHIJKLMNPQRSTVWXYZ*" (= AAPosInMatrix of matrixScores.go:117, which IS pinned);
 *       - BLOSUM62: the NCBI matrix; its gap row / column is taken as 0 here because align.go:127 recognises a gap
 *         feature by Score() == -GapOpen for ANY gap length and charges the extensions itself -- with a non-zero gap
 *         column a gap of two or more positions would not compare equal.  `gap_col` changes it;
 *       - ties (first maximum among diag, up, left; the first best cell in row-major order) and the way the traceback
 *         leaves a layer are this file's choice.  No test or fixture of the reference pins them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- matrixScores.go:22-105: "<matrix>_<open>_<extend>" -> lambda, K (values after diamond's score_matrix.cpp) ---- */
typedef struct { const char *key; double lambda, k; } ko_stat;
static const ko_stat KO_STATS[] = {
    {"blosum45_13_3", 0.207, 0.049}, {"blosum45_12_3", 0.199, 0.039}, {"blosum45_11_3", 0.190, 0.031}, {"blosum45_10_3", 0.179, 0.023},
    {"blosum45_16_2", 0.210, 0.051}, {"blosum45_15_2", 0.203, 0.041}, {"blosum45_14_2", 0.195, 0.032}, {"blosum45_13_2", 0.185, 0.024},
    {"blosum45_12_2", 0.171, 0.016}, {"blosum45_19_1", 0.205, 0.040}, {"blosum45_18_1", 0.198, 0.032}, {"blosum45_17_1", 0.189, 0.024},
    {"blosum45_16_1", 0.176, 0.016},
    {"blosum50_13_3", 0.212, 0.063}, {"blosum50_12_3", 0.206, 0.055}, {"blosum50_11_3", 0.197, 0.042}, {"blosum50_10_3", 0.186, 0.031},
    {"blosum50_9_3", 0.172, 0.022}, {"blosum50_16_2", 0.215, 0.066}, {"blosum50_15_2", 0.210, 0.058}, {"blosum50_14_2", 0.202, 0.045},
    {"blosum50_13_2", 0.193, 0.035}, {"blosum50_12_2", 0.181, 0.025}, {"blosum50_19_1", 0.212, 0.057}, {"blosum50_18_1", 0.207, 0.050},
    {"blosum50_17_1", 0.198, 0.037}, {"blosum50_16_1", 0.186, 0.025}, {"blosum50_15_1", 0.171, 0.015},
    {"blosum62_11_2", 0.297, 0.082}, {"blosum62_10_2", 0.291, 0.075}, {"blosum62_9_2", 0.279, 0.058}, {"blosum62_8_2", 0.264, 0.045},
    {"blosum62_7_2", 0.239, 0.027}, {"blosum62_6_2", 0.201, 0.012}, {"blosum62_13_1", 0.292, 0.071}, {"blosum62_12_1", 0.283, 0.059},
    {"blosum62_11_1", 0.267, 0.041}, {"blosum62_10_1", 0.243, 0.024}, {"blosum62_9_1", 0.206, 0.010},
    {"blosum80_25_2", 0.342, 0.17}, {"blosum80_13_2", 0.336, 0.15}, {"blosum80_9_2", 0.319, 0.11}, {"blosum80_8_2", 0.308, 0.090},
    {"blosum80_7_2", 0.293, 0.070}, {"blosum80_6_2", 0.268, 0.045}, {"blosum80_11_1", 0.314, 0.095}, {"blosum80_10_1", 0.299, 0.071},
    {"blosum80_9_1", 0.279, 0.048},
    {"blosum90_9_2", 0.310, 0.12}, {"blosum90_8_2", 0.300, 0.099}, {"blosum90_7_2", 0.283, 0.072}, {"blosum90_6_2", 0.259, 0.048},
    {"blosum90_11_1", 0.302, 0.093}, {"blosum90_10_1", 0.290, 0.075}, {"blosum90_9_1", 0.265, 0.044},
    {"pam250_15_3", 0.205, 0.049}, {"pam250_14_3", 0.200, 0.043}, {"pam250_13_3", 0.194, 0.036}, {"pam250_12_3", 0.186, 0.029},
    {"pam250_11_3", 0.174, 0.020}, {"pam250_17_2", 0.204, 0.047}, {"pam250_16_2", 0.198, 0.038}, {"pam250_15_2", 0.191, 0.031},
    {"pam250_14_2", 0.182, 0.024}, {"pam250_13_2", 0.171, 0.017}, {"pam250_21_1", 0.205, 0.045}, {"pam250_20_1", 0.199, 0.037},
    {"pam250_19_1", 0.192, 0.029}, {"pam250_18_1", 0.183, 0.021}, {"pam250_17_1", 0.171, 0.014},
    {"pam30_7_2", 0.305, 0.15}, {"pam30_6_2", 0.287, 0.11}, {"pam30_5_2", 0.264, 0.079}, {"pam30_10_1", 0.309, 0.15},
    {"pam30_9_1", 0.294, 0.11}, {"pam30_8_1", 0.270, 0.072},
    {"pam70_8_2", 0.301, 0.12}, {"pam70_7_2", 0.286, 0.093}, {"pam70_6_2", 0.264, 0.064}, {"pam70_11_1", 0.305, 0.12},
    {"pam70_10_1", 0.291, 0.091}, {"pam70_9_1", 0.270, 0.060},
};

int ko_matrix_scores(const char *sub_matrix, int gap_open, int gap_extend, double *lambda, double *k)
{
    char key[96];
    size_t n = 0;
    for (; sub_matrix[n] && n < 60; n++) key[n] = (char)((sub_matrix[n] >= 'A' && sub_matrix[n] <= 'Z') ? sub_matrix[n] + 32 : sub_matrix[n]);  /* strings.ToLower */
    /* fmt.Sprintf("%s_%d_%d") */
    int w = 0;
    {
        char num[32];
        int v[2] = { gap_open, gap_extend };
        for (int t = 0; t < 2; t++) {
            key[n++] = '_';
            long long x = v[t];
            if (x < 0) { key[n++] = '-'; x = -x; }
            w = 0;
            do { num[w++] = (char)('0' + x % 10); x /= 10; } while (x);
            while (w) key[n++] = num[--w];
        }
        key[n] = 0;
    }
    for (size_t i = 0; i < sizeof KO_STATS / sizeof KO_STATS[0]; i++)
        if (!strcmp(KO_STATS[i].key, key)) { *lambda = KO_STATS[i].lambda; *k = KO_STATS[i].k; return 0; }
    return 1;  /* "No matrix found" */
}

/* ---- biogo's protein alphabet = AAPosInMatrix (matrixScores.go:117): index of a letter, -1 outside ---- */
static const char KO_ALPHA[] = "-ABCDEFGHIJKLMNPQRSTVWXYZ*";
static int ko_letter(int c)   /* AAPosInMatrix: exact letters only (a Go map miss reads as 0: callers decide) */
{
    const char *p = c ? strchr(KO_ALPHA, c) : NULL;
    return p ? (int)(p - KO_ALPHA) : -1;
}
/* the aligner's index: biogo's alphabet.Protein is NOT case sensitive (NewGeneric(..., caseSensitive = false)) */
static int ko_dp_letter(int c)
{
    return ko_letter((c >= 'a' && c <= 'z') ? c - 32 : c);
}

/* NCBI BLOSUM62, in the NCBI file's order (public domain); J after the NCBI matrices that carry it */
static const char KO_NCBI_ORDER[] = "ARNDCQEGHILKMFPSTWYVBZX*";
static const signed char KO_B62[24][24] = {
    { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0,-4},
    {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1,-4},
    {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1,-4},
    {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2,-4},
    {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1,-4},
    {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1,-4},
    {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1,-4},
    {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1,-4},
    {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1,-4},
    {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1,-4},
    {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1,-4},
    {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1,-4},
    {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2,-4},
    { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0,-4},
    { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0,-4},
    {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2,-4},
    {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1,-4},
    { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1,-4},
    {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1,-4},
    {-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1},
};
/* J against ARNDCQEGHILKMFPSTWYVBZX* (NCBI matrices with J), J-J = 3 */
static const signed char KO_B62_J[24] = {-1,-2,-3,-3,-1,-2,-3,-4,-3, 3, 3,-3, 2, 0,-3,-2,-1,-2,-1, 2,-3,-3,-1,-4};

/* matrix.BLOSUM62 in the alphabet's order; row / column 0 = the gap column (see the header) */
int ko_b62(int a, int b, int gap_col)
{
    if (a < 0 || b < 0 || a > 25 || b > 25) return 0;
    if (a == 0 || b == 0) return (a == 0 && b == 0) ? 0 : gap_col;
    const char ca = KO_ALPHA[a], cb = KO_ALPHA[b];
    if (ca == 'J' && cb == 'J') return 3;
    if (ca == 'J' || cb == 'J') {
        const char o = ca == 'J' ? cb : ca;
        return KO_B62_J[strchr(KO_NCBI_ORDER, o) - KO_NCBI_ORDER];
    }
    return KO_B62[strchr(KO_NCBI_ORDER, ca) - KO_NCBI_ORDER][strchr(KO_NCBI_ORDER, cb) - KO_NCBI_ORDER];
}

typedef struct {
    float identity, similarity;
    int32_t length, mismatches, gap_openings, raw;
    double bitscore, evalue;
    int32_t q_start, q_end, s_start, s_end;
} ko_alignment;

enum { L_DIAG = 0, L_UP = 1, L_LEFT = 2 };

/* first maximum of three (biogo's max(&[3]int) walks the array in order) */
static int ko_arg3(const int s[3])
{
    int b = 0;
    if (s[1] > s[b]) b = 1;
    if (s[2] > s[b]) b = 2;
    return b;
}

/*
 * align.Align (align.go:46-161).  aln_out receives the three lines of AlnString (query row, match row, subject row),
 * each *aln_len characters; it must hold 3 * (nq + ns) bytes.  Returns 0, 1 = "No matrix found" (align.go:50-52: the
 * caller keeps an empty AlignmentResult), 2 = a letter outside the alphabet (SWAffine.Align fails; the reference ignores
 * the error and formats an empty alignment: reported as a failed pair instead), 3 = out of memory.
 */
int ko_align(const char *query, int nq, const char *subject, int ns, uint64_t number_of_aa, const char *sub_matrix,
             int gap_open, int gap_extend, int dp_gap_open, int gap_col, ko_alignment *out, char *aln_out, int *aln_len)
{
    double lambda, kk;
    memset(out, 0, sizeof *out);
    *aln_len = 0;
    if (ko_matrix_scores(sub_matrix, gap_open, gap_extend, &lambda, &kk)) return 1;
    /* the similarity marks use matrixScores.SubMatrix (align.go:96); only BLOSUM62's data is restated here */
    if (strncmp(sub_matrix, "blosum62", 8) && strncmp(sub_matrix, "BLOSUM62", 8)) return 1;
    char *a = (char *)malloc((size_t)nq + 1), *b = (char *)malloc((size_t)ns + 1);
    if (!a || !b) { free(a); free(b); return 3; }
    for (int i = 0; i < nq; i++) a[i] = (query[i] == 'u' || query[i] == 'U') ? '*' : query[i];       /* align.go:54 */
    for (int j = 0; j < ns; j++) b[j] = (subject[j] == 'u' || subject[j] == 'U') ? '*' : subject[j]; /* align.go:55 */
    for (int i = 0; i < nq; i++) if (ko_dp_letter(a[i]) < 0) { free(a); free(b); return 2; }
    for (int j = 0; j < ns; j++) if (ko_dp_letter(b[j]) < 0) { free(a); free(b); return 2; }
    const int r = nq + 1, c = ns + 1;
    int (*tab)[3] = (int (*)[3])calloc((size_t)r * c, sizeof(int[3]));
    unsigned char *from = (unsigned char *)calloc((size_t)r * c, 1);  /* 2 bits per layer: predecessor layer + 1, 0 = start */
    if (!tab || !from) { free(a); free(b); free(tab); free(from); return 3; }
    int max_s = 0, max_i = 0, max_j = 0, max_l = L_DIAG;
    for (int i = 1; i < r; i++) {
        const int rv = ko_dp_letter(a[i - 1]);
        for (int j = 1; j < c; j++) {
            const int qv = ko_dp_letter(b[j - 1]);
            const size_t p = (size_t)i * c + j;
            int s[3], k;
            unsigned f = 0;
            /* diag: the best of the three layers at (i-1, j-1) + the substitution score, floored at zero */
            s[0] = tab[p - c - 1][L_DIAG]; s[1] = tab[p - c - 1][L_UP]; s[2] = tab[p - c - 1][L_LEFT];
            k = ko_arg3(s);
            int v = s[k] + ko_b62(rv, qv, gap_col);
            if (v > 0) { tab[p][L_DIAG] = v; f |= (unsigned)(s[k] > 0 ? k + 1 : 0); }
            /* up: a gap position that consumes query[i-1]; entering the layer costs the gap-open penalty */
            s[0] = tab[p - c][L_DIAG] + dp_gap_open + ko_b62(rv, 0, gap_col);
            s[1] = tab[p - c][L_UP] + ko_b62(rv, 0, gap_col);
            s[2] = tab[p - c][L_LEFT] + dp_gap_open + ko_b62(rv, 0, gap_col);
            k = ko_arg3(s);
            if (s[k] > 0) { tab[p][L_UP] = s[k]; f |= (unsigned)(k + 1) << 2; }
            /* left: a gap position that consumes subject[j-1] */
            s[0] = tab[p - 1][L_DIAG] + dp_gap_open + ko_b62(0, qv, gap_col);
            s[1] = tab[p - 1][L_UP] + dp_gap_open + ko_b62(0, qv, gap_col);
            s[2] = tab[p - 1][L_LEFT] + ko_b62(0, qv, gap_col);
            k = ko_arg3(s);
            if (s[k] > 0) { tab[p][L_LEFT] = s[k]; f |= (unsigned)(k + 1) << 4; }
            from[p] = (unsigned char)f;
            for (int l = 0; l < 3; l++)
                if (tab[p][l] > max_s) { max_s = tab[p][l]; max_i = i; max_j = j; max_l = l; }
        }
    }
    /* traceback -> operations in reverse: 'M' (both), 'U' (query letter against '-'), 'L' ('-' against subject letter) */
    char *ops = (char *)malloc((size_t)nq + ns + 2);
    if (!ops) { free(a); free(b); free(tab); free(from); return 3; }
    int n_ops = 0, i = max_i, j = max_j, l = max_l;
    while (i > 0 && j > 0 && max_s > 0) {
        const size_t p = (size_t)i * c + j;
        if (tab[p][l] <= 0) break;
        const unsigned pred = (from[p] >> (2 * l)) & 3u;
        ops[n_ops++] = l == L_DIAG ? 'M' : l == L_UP ? 'U' : 'L';
        if (l == L_DIAG) { i--; j--; } else if (l == L_UP) i--; else j--;
        if (pred == 0) break;   /* the alignment starts here */
        l = (int)pred - 1;
    }
    const int qs = i, ss = j;   /* 0-based start of the aligned ranges */
    /* align.Format + align.go:70-103, walking the operations forward */
    float identity = 0, similarity = 0, nb_pos = 0;
    int mismatches = 0, len = n_ops;
    char *row_a = aln_out, *row_m = aln_out + len, *row_b = aln_out + 2 * len;
    int ai = qs, bj = ss;
    for (int t = 0; t < n_ops; t++) {
        const char op = ops[n_ops - 1 - t];
        const char ca = op == 'L' ? '-' : a[ai++], cb = op == 'U' ? '-' : b[bj++];
        row_a[t] = ca; row_b[t] = cb;
        if (cb == ca) { identity += 1; similarity += 1; row_m[t] = cb; }
        else {
            if (cb != '-' && ca != '-') mismatches++;
            /* GetAlnScoreAA(matrixScores, b, a) > 0 with AAPosInMatrix (a map miss reads as index 0) */
            const int ib = ko_letter(cb) < 0 ? 0 : ko_letter(cb), ia = ko_letter(ca) < 0 ? 0 : ko_letter(ca);
            if (ko_b62(ib, ia, gap_col) > 0) { similarity += 1; row_m[t] = '+'; } else row_m[t] = ' ';
        }
        nb_pos += 1;
    }
    identity = (identity / nb_pos) * 100;      /* (0 / 0 = NaN for an empty alignment, as in the reference) */
    similarity = (similarity / nb_pos) * 100;
    /* align.go:105-133: the feature pairs = maximal runs of one operation; a match run scores the sum of its substitution
       scores, a gap run scores the gap-open penalty (see the header) */
    int raw = 0, gap_openings = 0;
    ai = qs; bj = ss;
    for (int t = 0; t < n_ops;) {
        const char op = ops[n_ops - 1 - t];
        int run = 0, score = 0;
        while (t + run < n_ops && ops[n_ops - 1 - t - run] == op) {
            if (op == 'M') score += ko_b62(ko_dp_letter(a[ai]), ko_dp_letter(b[bj]), gap_col);
            else if (op == 'U') score += ko_b62(ko_dp_letter(a[ai]), 0, gap_col);
            else score += ko_b62(0, ko_dp_letter(b[bj]), gap_col);
            if (op != 'L') ai++;
            if (op != 'U') bj++;
            run++;
        }
        if (op != 'M') score += dp_gap_open;
        raw += score;
        if (score == -gap_open) {              /* align.go:127 */
            gap_openings += 1;
            raw -= (run - 1) * gap_extend;     /* gapLen = max of the two feature lengths = the run */
        }
        t += run;
    }
    out->identity = identity;
    out->similarity = similarity;
    out->length = len;
    out->mismatches = mismatches;
    out->gap_openings = gap_openings;
    out->raw = raw;
    out->bitscore = ((lambda * (double)raw) - log(kk)) / log(2);
    out->evalue = (double)nq * (double)number_of_aa / pow(2, out->bitscore);
    out->q_start = n_ops ? qs + 1 : 1;         /* queryStart + 1 (zero features: 0 + 1) */
    out->q_end = n_ops ? max_i : 0;
    out->s_start = n_ops ? ss + 1 : 1;
    out->s_end = n_ops ? max_j : 0;
    *aln_len = len;
    free(a); free(b); free(tab); free(from); free(ops);
    return 0;
}
