// align.hip — the alignment step of the reference (`-aln`), batched on the device.
//
// Replaces, for all reported hits of a batch at once, the loop of QueryResultHandler (pkg/search/search.go:483-494):
//     alignment, err := align.Align(query.Sequence, hitEntry.Sequence, dbStats, SubMatrix, GapOpen, GapExtend)
// i.e. pkg/align/align.go:46-161 -- a local alignment with affine gaps of every (query, hit) pair, then identity,
// similarity, mismatches, gap openings, raw score, bit score, e-value, coordinates and the three-line alignment
// string.  The reference calls github.com/biogo/biogo v1.0.1's align.SWAffine{Matrix: matrix.BLOSUM62, GapOpen: -11}
// (align.go:62-67, fixed whatever the options say); that library is not part of the reference tree, so the recurrence
// below restates its published algorithm and the choices that library makes on ties are this file's own (DESIGN.md
// section 7: which of the outputs the reference's files pin, and which they do not).  The arithmetic around it -- the
// (matrix, open, extend) -> (lambda, K) table, BitScore, EValue, the float32 identity / similarity, the raw score
// corrected by (gap length - 1) * GapExtend per gap, the +1 on the start coordinates -- follows align.go and
// matrixScores.go line by line.
//
// Device work (a 10 000-query batch with MaxResults 10 is 100 000 pairs of ~350 x 350 cells = 1.2e10 cells):
//   align_wave_kernel   one WAVE per pair, the classic anti-diagonal wavefront: the 64 lanes hold 64 consecutive query
//                       rows, step d computes cell (row, d - lane); the three layers of the cell above come from the
//                       neighbouring lane's registers (one DPP wave shift each), those of the diagonal are last step's
//                       "above", those of the left cell the lane's own last results -- no memory traffic for the
//                       recurrence at all.  Strips of 64 rows hand their last row to the next strip through LDS, the
//                       subject's letter indices and the 26 x 26 matrix sit in LDS too.  One byte per cell (the
//                       predecessor layer of each of the three layers) goes to HBM diagonal-major, 64 bytes per step
//                       and wave, coalesced; lane 0 walks it backwards for the traceback.
//   align_kernel        one LANE per pair with the rows in HBM: the fallback for subjects longer than the LDS row
//                       buffer holds (ALN_WAVE_NS letters).
// Integer arithmetic only; the host turns the operations into the reference's strings and numbers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kaamer_internal.h"

namespace {

// ---- matrixScores.go:22-105 ("<matrix>_<open>_<extend>" -> lambda, K; values after diamond's score_matrix.cpp) ----
struct StatRow { const char *matrix; int open, ext; double lambda, k; };
const StatRow STATS[] = {
    {"blosum45", 13, 3, 0.207, 0.049}, {"blosum45", 12, 3, 0.199, 0.039}, {"blosum45", 11, 3, 0.190, 0.031}, {"blosum45", 10, 3, 0.179, 0.023},
    {"blosum45", 16, 2, 0.210, 0.051}, {"blosum45", 15, 2, 0.203, 0.041}, {"blosum45", 14, 2, 0.195, 0.032}, {"blosum45", 13, 2, 0.185, 0.024},
    {"blosum45", 12, 2, 0.171, 0.016}, {"blosum45", 19, 1, 0.205, 0.040}, {"blosum45", 18, 1, 0.198, 0.032}, {"blosum45", 17, 1, 0.189, 0.024},
    {"blosum45", 16, 1, 0.176, 0.016},
    {"blosum50", 13, 3, 0.212, 0.063}, {"blosum50", 12, 3, 0.206, 0.055}, {"blosum50", 11, 3, 0.197, 0.042}, {"blosum50", 10, 3, 0.186, 0.031},
    {"blosum50", 9, 3, 0.172, 0.022}, {"blosum50", 16, 2, 0.215, 0.066}, {"blosum50", 15, 2, 0.210, 0.058}, {"blosum50", 14, 2, 0.202, 0.045},
    {"blosum50", 13, 2, 0.193, 0.035}, {"blosum50", 12, 2, 0.181, 0.025}, {"blosum50", 19, 1, 0.212, 0.057}, {"blosum50", 18, 1, 0.207, 0.050},
    {"blosum50", 17, 1, 0.198, 0.037}, {"blosum50", 16, 1, 0.186, 0.025}, {"blosum50", 15, 1, 0.171, 0.015},
    {"blosum62", 11, 2, 0.297, 0.082}, {"blosum62", 10, 2, 0.291, 0.075}, {"blosum62", 9, 2, 0.279, 0.058}, {"blosum62", 8, 2, 0.264, 0.045},
    {"blosum62", 7, 2, 0.239, 0.027}, {"blosum62", 6, 2, 0.201, 0.012}, {"blosum62", 13, 1, 0.292, 0.071}, {"blosum62", 12, 1, 0.283, 0.059},
    {"blosum62", 11, 1, 0.267, 0.041}, {"blosum62", 10, 1, 0.243, 0.024}, {"blosum62", 9, 1, 0.206, 0.010},
    {"blosum80", 25, 2, 0.342, 0.17}, {"blosum80", 13, 2, 0.336, 0.15}, {"blosum80", 9, 2, 0.319, 0.11}, {"blosum80", 8, 2, 0.308, 0.090},
    {"blosum80", 7, 2, 0.293, 0.070}, {"blosum80", 6, 2, 0.268, 0.045}, {"blosum80", 11, 1, 0.314, 0.095}, {"blosum80", 10, 1, 0.299, 0.071},
    {"blosum80", 9, 1, 0.279, 0.048},
    {"blosum90", 9, 2, 0.310, 0.12}, {"blosum90", 8, 2, 0.300, 0.099}, {"blosum90", 7, 2, 0.283, 0.072}, {"blosum90", 6, 2, 0.259, 0.048},
    {"blosum90", 11, 1, 0.302, 0.093}, {"blosum90", 10, 1, 0.290, 0.075}, {"blosum90", 9, 1, 0.265, 0.044},
    {"pam250", 15, 3, 0.205, 0.049}, {"pam250", 14, 3, 0.200, 0.043}, {"pam250", 13, 3, 0.194, 0.036}, {"pam250", 12, 3, 0.186, 0.029},
    {"pam250", 11, 3, 0.174, 0.020}, {"pam250", 17, 2, 0.204, 0.047}, {"pam250", 16, 2, 0.198, 0.038}, {"pam250", 15, 2, 0.191, 0.031},
    {"pam250", 14, 2, 0.182, 0.024}, {"pam250", 13, 2, 0.171, 0.017}, {"pam250", 21, 1, 0.205, 0.045}, {"pam250", 20, 1, 0.199, 0.037},
    {"pam250", 19, 1, 0.192, 0.029}, {"pam250", 18, 1, 0.183, 0.021}, {"pam250", 17, 1, 0.171, 0.014},
    {"pam30", 7, 2, 0.305, 0.15}, {"pam30", 6, 2, 0.287, 0.11}, {"pam30", 5, 2, 0.264, 0.079}, {"pam30", 10, 1, 0.309, 0.15},
    {"pam30", 9, 1, 0.294, 0.11}, {"pam30", 8, 1, 0.270, 0.072},
    {"pam70", 8, 2, 0.301, 0.12}, {"pam70", 7, 2, 0.286, 0.093}, {"pam70", 6, 2, 0.264, 0.064}, {"pam70", 11, 1, 0.305, 0.12},
    {"pam70", 10, 1, 0.291, 0.091}, {"pam70", 9, 1, 0.270, 0.060},
};

// GetMatrixScores (matrixScores.go:107-115): the key is strings.ToLower(subMatrix) + "_" + open + "_" + extend
bool matrix_scores(const char *sub_matrix, int gap_open, int gap_extend, double *lambda, double *k)
{
    std::string m(sub_matrix ? sub_matrix : "");
    for (char &c : m) if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    for (const StatRow &r : STATS)
        if (m == r.matrix && gap_open == r.open && gap_extend == r.ext) { *lambda = r.lambda; *k = r.k; return true; }
    return false;
}

// AAPosInMatrix (matrixScores.go:117) = biogo's protein alphabet
const char ALPHA[] = "-ABCDEFGHIJKLMNPQRSTVWXYZ*";
constexpr int NL = 26;

int letter_index(int c)   // exact letters (the map of matrixScores.go:117); -1: a map miss
{
    if (c == '-') return 0;
    if (c == '*') return 25;
    if (c >= 'A' && c <= 'Z' && c != 'O' && c != 'U') {
        int i = c - 'A' + 1;          // A=1 .. N=14
        if (c > 'O') i--;             // P=15 .. T=19
        if (c > 'U') i--;             // V=20 .. Z=24
        return i;
    }
    return -1;
}

// BLOSUM62 in the alphabet's order.  Rows of the NCBI matrix (ARNDCQEGHILKMFPSTWYVBZX*), re-indexed once at start-up;
// J after the NCBI matrices that carry it; row / column 0 (the gap) = gap_col.  It is 0: align.go:127 recognises a gap feature
// by Score() == -GapOpen for ANY gap length and charges the extensions itself (align.go:129-130), which holds only when a gap
// position beyond the first costs nothing in the aligner's own score
const char NCBI[] = "ARNDCQEGHILKMFPSTWYVBZX*";
const signed char B62[24][24] = {
    { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0,-4}, {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1,-4},
    {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1,-4}, {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2,-4}, {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1,-4},
    {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4}, { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1,-4},
    {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1,-4}, {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1,-4},
    {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1,-4}, {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1,-4},
    {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1,-4}, {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1,-4},
    {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2,-4}, { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0,-4},
    { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0,-4}, {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2,-4},
    {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1,-4}, { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1,-4},
    {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1,-4}, {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1,-4}, {-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1},
};
const signed char B62_J[24] = {-1,-2,-3,-3,-1,-2,-3,-4,-3, 3, 3,-3, 2, 0,-3,-2,-1,-2,-1, 2,-3,-3,-1,-4};

void fill_matrix(int *m, int gap_col)
{
    for (int a = 0; a < NL; a++)
        for (int b = 0; b < NL; b++) {
            int v;
            if (a == 0 || b == 0) v = (a == 0 && b == 0) ? 0 : gap_col;
            else if (ALPHA[a] == 'J' && ALPHA[b] == 'J') v = 3;
            else if (ALPHA[a] == 'J' || ALPHA[b] == 'J') v = B62_J[strchr(NCBI, ALPHA[a] == 'J' ? ALPHA[b] : ALPHA[a]) - NCBI];
            else v = B62[strchr(NCBI, ALPHA[a]) - NCBI][strchr(NCBI, ALPHA[b]) - NCBI];
            m[a * NL + b] = v;
        }
}

struct PairDesc {   // one pair on the device
    uint32_t q_off, nq, s_off, ns;
};

struct WaveDesc {   // 64 pairs of similar size
    uint32_t first_pair, n_pairs;
    uint32_t max_nq, max_ns;
    uint64_t row_off;   // ints: 3 layers x (max_ns + 1) columns x 64 lanes
    uint64_t dir_off;   // bytes: max_nq x max_ns cells x 64 lanes
    uint64_t ops_off;   // bytes: (max_nq + max_ns) x 64 lanes
};

struct PairOut {
    int32_t max_s, end_i, end_j, start_i, start_j, n_ops;
};

struct AlignParams {
    const uint8_t *codes;      // letter indices of every sequence (U already '*', case folded)
    const PairDesc *pairs;
    const WaveDesc *waves;
    const int *matrix;         // 26 x 26
    int gap_open;              // the aligner's GapOpen (negative)
    int *rows;
    uint8_t *dirs;
    uint8_t *ops;
    PairOut *out;
};

// first maximum of (a, b, c): its value and its index + 1
__device__ __forceinline__ void arg3(int a, int b, int c, int &v, unsigned &k)
{
    v = a; k = 1u;
    if (b > v) { v = b; k = 2u; }
    if (c > v) { v = c; k = 3u; }
}

__global__ __launch_bounds__(64) void align_kernel(AlignParams p)
{
    __shared__ int s_m[NL * NL];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < NL * NL; i += 64) s_m[i] = p.matrix[i];
    __syncthreads();
    const WaveDesc w = p.waves[blockIdx.x];
    const bool live = lane < w.n_pairs;
    PairDesc d;
    d.q_off = d.nq = d.s_off = d.ns = 0;
    if (live) d = p.pairs[w.first_pair + lane];
    const uint32_t W = w.max_ns + 1;
    int *const row = p.rows + w.row_off + lane;   // layer l, column j: row[(l * W + j) * 64]
    uint8_t *const dir = p.dirs + w.dir_off + lane;
    // row 0: zeros
    for (uint32_t j = 0; j <= w.max_ns; j++) {
        row[(0 * (uint64_t)W + j) * 64] = 0;
        row[(1 * (uint64_t)W + j) * 64] = 0;
        row[(2 * (uint64_t)W + j) * 64] = 0;
    }
    int max_s = 0, max_i = 0, max_j = 0, max_l = 0;
    for (uint32_t i = 1; i <= w.max_nq; i++) {
        const bool row_live = live && i <= d.nq;
        const int rv = row_live ? (int)p.codes[d.q_off + i - 1] : 0;
        const int gr = s_m[rv * NL];                 // gap-column score of the query letter
        // column 0 of this row: zeros in every layer; the previous row's column 0 likewise
        int dg_m = 0, dg_u = 0, dg_l = 0;            // previous row, column j - 1
        int lf_m = 0, lf_u = 0, lf_l = 0;            // this row, column j - 1
        for (uint32_t j = 1; j <= w.max_ns; j++) {
            const int up_m = row[(0 * (uint64_t)W + j) * 64], up_u = row[(1 * (uint64_t)W + j) * 64], up_l = row[(2 * (uint64_t)W + j) * 64];
            const bool cell = row_live && j <= d.ns;
            const int qv = cell ? (int)p.codes[d.s_off + j - 1] : 0;
            int v, cm = 0, cu = 0, cl = 0;
            unsigned k, f = 0;
            // diag: the best layer of (i-1, j-1) + the substitution score, floored at zero
            arg3(dg_m, dg_u, dg_l, v, k);
            const int pm = v;
            v += s_m[rv * NL + qv];
            if (v > 0) { cm = v; f |= pm > 0 ? k : 0u; }
            // up: a gap position that consumes the query letter
            arg3(up_m + p.gap_open + gr, up_u + gr, up_l + p.gap_open + gr, v, k);
            if (v > 0) { cu = v; f |= k << 2; }
            // left: a gap position that consumes the subject letter
            const int gq = s_m[qv];
            arg3(lf_m + p.gap_open + gq, lf_u + p.gap_open + gq, lf_l + gq, v, k);
            if (v > 0) { cl = v; f |= k << 4; }
            if (!cell) { cm = cu = cl = 0; f = 0; }
            row[(0 * (uint64_t)W + j) * 64] = cm;
            row[(1 * (uint64_t)W + j) * 64] = cu;
            row[(2 * (uint64_t)W + j) * 64] = cl;
            dir[((uint64_t)(i - 1) * w.max_ns + (j - 1)) * 64] = (uint8_t)f;
            if (cm > max_s) { max_s = cm; max_i = (int)i; max_j = (int)j; max_l = 0; }
            if (cu > max_s) { max_s = cu; max_i = (int)i; max_j = (int)j; max_l = 1; }
            if (cl > max_s) { max_s = cl; max_i = (int)i; max_j = (int)j; max_l = 2; }
            dg_m = up_m; dg_u = up_u; dg_l = up_l;
            lf_m = cm; lf_u = cu; lf_l = cl;
        }
    }
    if (!live) return;
    // traceback: operations in reverse ('M' both letters, 'U' the query letter against a gap, 'L' a gap against the subject letter)
    uint8_t *const ops = p.ops + w.ops_off + lane;
    int i = max_i, j = max_j, l = max_l, n_ops = 0;
    while (i > 0 && j > 0 && max_s > 0) {
        const unsigned f = dir[((uint64_t)(i - 1) * w.max_ns + (j - 1)) * 64];
        const unsigned pred = (f >> (2 * l)) & 3u;
        ops[(uint64_t)n_ops * 64] = l == 0 ? 'M' : l == 1 ? 'U' : 'L';
        n_ops++;
        if (l == 0) { i--; j--; } else if (l == 1) i--; else j--;
        if (pred == 0) break;
        l = (int)pred - 1;
    }
    PairOut o;
    o.max_s = max_s; o.end_i = max_i; o.end_j = max_j; o.start_i = i; o.start_j = j; o.n_ops = n_ops;
    p.out[w.first_pair + lane] = o;
}

// ---- one wave per pair: anti-diagonal wavefront ----------------------------------------------------------------------
#define ALN_WAVE_NS 2048u   /* longest subject the LDS row buffer holds (3 layers x 4 bytes x (ALN_WAVE_NS + 1)) */

struct WPair {
    uint32_t q_off, nq, s_off, ns;
    uint64_t dir_off;   // bytes: strips x (ns + 63) steps x 64 lanes
    uint64_t ops_off;   // bytes: nq + ns
};

struct WaveParams {
    const uint8_t *codes;
    const WPair *pairs;
    const int *matrix;
    int gap_open;
    uint8_t *dirs;
    uint8_t *ops;
    PairOut *out;
    uint32_t first;     // index of pairs[0] in `out`
};

// value of the next lower lane (lane 0: `first`): DPP wave_shr:1
__device__ __forceinline__ int from_lower_lane(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

__global__ __launch_bounds__(64) void align_wave_kernel(WaveParams p)
{
    __shared__ int s_m[NL * NL];
    __shared__ int s_bnd[3][ALN_WAVE_NS + 1];   // the last row of the previous strip, per layer, by column (0 = zeros)
    __shared__ uint8_t s_sub[ALN_WAVE_NS];
    const uint32_t lane = threadIdx.x;
    const WPair d = p.pairs[blockIdx.x];
    for (uint32_t i = lane; i < NL * NL; i += 64) s_m[i] = p.matrix[i];
    for (uint32_t j = lane; j < d.ns; j += 64) s_sub[j] = p.codes[d.s_off + j];
    for (uint32_t j = lane; j <= d.ns; j += 64) { s_bnd[0][j] = 0; s_bnd[1][j] = 0; s_bnd[2][j] = 0; }
    __syncthreads();
    const uint32_t steps = d.ns + 63u;
    uint8_t *const dir = p.dirs + d.dir_off + lane;
    int best = 0, best_i = 0, best_j = 0, best_l = 0;
    uint32_t strip = 0;
    for (uint32_t i0 = 0; i0 < d.nq; i0 += 64, strip++) {
        const uint32_t i = i0 + lane + 1;                 // this lane's row (1-based)
        const bool row_live = i <= d.nq;
        const int rv = row_live ? (int)p.codes[d.q_off + i - 1] : 0;
        const int gr = s_m[rv * NL];                      // the gap-column score of the query letter
        int cm = 0, cu = 0, cl = 0;                       // this lane's last cell (zeros: column 0 / not a cell)
        int pu_m = 0, pu_u = 0, pu_l = 0;                 // the cell above last step's cell = this step's diagonal
        uint8_t *const sdir = dir + (uint64_t)strip * steps * 64;
        for (uint32_t st = 0; st < steps; st++) {
            const int j = (int)st - (int)lane + 1;        // this step's column (1-based)
            const bool cell = row_live && j >= 1 && j <= (int)d.ns;
            // the cell above: the lower lane's last results (its row is i - 1, its last column was j); lane 0 reads the
            // previous strip's last row
            const int jc = j < 0 ? 0 : (j > (int)d.ns ? (int)d.ns : j);
            const int b_m = s_bnd[0][jc], b_u = s_bnd[1][jc], b_l = s_bnd[2][jc];
            const int up_m = from_lower_lane(cm, b_m), up_u = from_lower_lane(cu, b_u), up_l = from_lower_lane(cl, b_l);
            const int qv = cell ? (int)s_sub[j - 1] : 0;
            int v, nm = 0, nu = 0, nl = 0;
            unsigned k, f = 0;
            arg3(pu_m, pu_u, pu_l, v, k);                 // diag: the best layer of (i-1, j-1) + the substitution score
            const int pm = v;
            v += s_m[rv * NL + qv];
            if (v > 0) { nm = v; f |= pm > 0 ? k : 0u; }
            arg3(up_m + p.gap_open + gr, up_u + gr, up_l + p.gap_open + gr, v, k);       // up: consumes the query letter
            if (v > 0) { nu = v; f |= k << 2; }
            const int gq = s_m[qv];
            arg3(cm + p.gap_open + gq, cu + p.gap_open + gq, cl + gq, v, k);             // left: consumes the subject letter
            if (v > 0) { nl = v; f |= k << 4; }
            if (!cell) { nm = nu = nl = 0; f = 0; }
            sdir[(uint64_t)st * 64] = (uint8_t)f;
            // the first best cell in row-major order: within a row columns ascend, a lane's rows ascend with the strips
            if (nm > best) { best = nm; best_i = (int)i; best_j = j; best_l = 0; }
            if (nu > best) { best = nu; best_i = (int)i; best_j = j; best_l = 1; }
            if (nl > best) { best = nl; best_i = (int)i; best_j = j; best_l = 2; }
            pu_m = up_m; pu_u = up_u; pu_l = up_l;
            cm = nm; cu = nu; cl = nl;
            // the strip's last row feeds the next strip (column j was read by lane 0 sixty-three steps ago)
            if (lane == 63 && cell) { s_bnd[0][j] = nm; s_bnd[1][j] = nu; s_bnd[2][j] = nl; }
        }
        __syncthreads();   // (one wave: orders the LDS row between strips)
    }
    // the best cell over the lanes: highest score, then the smallest row
    unsigned long long key = ((unsigned long long)(uint32_t)best << 32) | (uint32_t)(0x7FFFFFFF - best_i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
    }
    const int w_best = (int)(key >> 32), w_i = 0x7FFFFFFF - (int)(uint32_t)key;
    const unsigned long long mine = __ballot(best == w_best && best_i == w_i && w_best > 0);
    int max_j = 0, max_l = 0;
    if (mine) {
        const int src = __ffsll((long long)mine) - 1;
        max_j = __shfl(best_j, src, 64);
        max_l = __shfl(best_l, src, 64);
    }
    __threadfence_block();
    __syncthreads();
    if (lane != 0) return;
    // traceback (lane 0): cell (i, j) lives in strip (i-1)/64 at step (j-1) + (i-1)%64, lane (i-1)%64
    uint8_t *const ops = p.ops + d.ops_off;
    int i = w_best > 0 ? w_i : 0, j = max_j, l = max_l, n_ops = 0;
    const int end_i = i, end_j = j;
    while (i > 0 && j > 0) {
        const uint32_t ln = (uint32_t)(i - 1) & 63u, sp = (uint32_t)(i - 1) >> 6;
        const unsigned f = p.dirs[d.dir_off + ((uint64_t)sp * steps + (uint32_t)(j - 1) + ln) * 64 + ln];
        const unsigned pred = (f >> (2 * l)) & 3u;
        ops[n_ops++] = l == 0 ? 'M' : l == 1 ? 'U' : 'L';
        if (l == 0) { i--; j--; } else if (l == 1) i--; else j--;
        if (pred == 0) break;
        l = (int)pred - 1;
    }
    PairOut o;
    o.max_s = w_best; o.end_i = end_i; o.end_j = end_j; o.start_i = i; o.start_j = j; o.n_ops = n_ops;
    p.out[p.first + blockIdx.x] = o;
}

template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        if (hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) != hipSuccess) { p = nullptr; return kaamer_fail(KAAMER_E_NOMEM, "align: device allocation of %zu bytes", n * sizeof(T)); }
        return KAAMER_OK;
    }
};

}  // namespace

struct kaamer_alignments {
    std::vector<kaamer_alignment> items;
    std::vector<char> text;
};

#define ALN_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return kaamer_fail(KAAMER_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int kaamer_align_matrix_scores(const char *sub_matrix, int32_t gap_open, int32_t gap_extend, double *lambda, double *k)
{
    if (!sub_matrix || !lambda || !k) return kaamer_fail(KAAMER_E_ARG, "align_matrix_scores: bad argument");
    if (!matrix_scores(sub_matrix, gap_open, gap_extend, lambda, k)) return kaamer_fail(KAAMER_E_ARG, "No matrix found");
    return KAAMER_OK;
}

int32_t kaamer_align_matrix_entry(int32_t a, int32_t b)
{
    static int m[NL * NL];
    static bool done = false;
    if (!done) { fill_matrix(m, 0); done = true; }
    const int ia = letter_index(a), ib = letter_index(b);
    return m[(ia < 0 ? 0 : ia) * NL + (ib < 0 ? 0 : ib)];   // (a map miss reads as index 0, as in GetAlnScoreAA)
}

uint32_t kaamer_alignments_count(const kaamer_alignments *a) { return a ? (uint32_t)a->items.size() : 0u; }
const kaamer_alignment *kaamer_alignments_items(const kaamer_alignments *a) { return a ? a->items.data() : nullptr; }
const char *kaamer_alignments_text(const kaamer_alignments *a) { return a ? a->text.data() : nullptr; }
void kaamer_alignments_free(kaamer_alignments *a) { delete a; }

int kaamer_align_pairs(int device, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs, const uint32_t *pair_query,
                       const uint32_t *pair_subject, uint32_t n_pairs, uint64_t number_of_aa, const char *sub_matrix, int32_t gap_open,
                       int32_t gap_extend, kaamer_alignments **out)
{
    if (!out || !offsets || (!seqs && n_seqs && offsets[n_seqs]) || (n_pairs && (!pair_query || !pair_subject)) || !sub_matrix)
        return kaamer_fail(KAAMER_E_ARG, "align_pairs: bad argument");
    *out = nullptr;
    for (uint32_t i = 0; i < n_pairs; i++)
        if (pair_query[i] >= n_seqs || pair_subject[i] >= n_seqs) return kaamer_fail(KAAMER_E_ARG, "align_pairs: pair %u names a sequence that is not there", i);
    kaamer_alignments *res = new (std::nothrow) kaamer_alignments();
    if (!res) return kaamer_fail(KAAMER_E_NOMEM, "alignments");
    res->items.assign(n_pairs, kaamer_alignment());
    for (kaamer_alignment &a : res->items) memset(&a, 0, sizeof a);
    double lambda = 0, kk = 0;
    // align.go:48-52: GetMatrixScores fails -> an empty AlignmentResult and an error (the caller `continue`s).  The similarity
    // marks read matrixScores.SubMatrix (align.go:96): only BLOSUM62's data is carried here, other matrices report the same
    std::string mlow(sub_matrix);
    for (char &c : mlow) if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    if (!matrix_scores(sub_matrix, gap_open, gap_extend, &lambda, &kk) || mlow != "blosum62") {
        for (kaamer_alignment &a : res->items) a.status = 1;
        *out = res;
        return KAAMER_OK;
    }
    int gap_col = 0, dp_open = -11;                     // align.go:62-65: BLOSUM62, GapOpen -11, whatever the options are
    if (const char *e = getenv("KAAMER_ALIGN_GAP_COLUMN")) gap_col = atoi(e);   // biogo's gap column, should it differ (see the header)
    int matrix[NL * NL];
    fill_matrix(matrix, gap_col);
    // ---- letters: [uU] -> '*' (align.go:54-55), then the aligner's index (not case sensitive); a letter outside: the pair fails
    const uint64_t total = n_seqs ? offsets[n_seqs] : 0;
    std::vector<uint8_t> codes((size_t)total + 1, 0);
    std::vector<uint8_t> seq_bad(n_seqs, 0);
    for (uint32_t s = 0; s < n_seqs; s++)
        for (uint64_t i = offsets[s]; i < offsets[s + 1]; i++) {
            int c = seqs[i];
            if (c == 'u' || c == 'U') c = '*';
            const int idx = letter_index((c >= 'a' && c <= 'z') ? c - 32 : c);
            if (idx < 0) seq_bad[s] = 1;
            codes[(size_t)i] = (uint8_t)(idx < 0 ? 0 : idx);
        }
    // ---- the pairs the device takes, sorted by size so that the 64 lanes of a wave walk similar matrices
    struct Work { uint32_t pair; uint32_t nq, ns; };
    std::vector<Work> work;
    for (uint32_t i = 0; i < n_pairs; i++) {
        const uint32_t q = pair_query[i], s = pair_subject[i];
        if (seq_bad[q] || seq_bad[s]) { res->items[i].status = 2; continue; }
        const uint64_t nq = offsets[q + 1] - offsets[q], ns = offsets[s + 1] - offsets[s];
        if (nq > 0x3FFFFFFFull || ns > 0x3FFFFFFFull) { res->items[i].status = 3; continue; }
        work.push_back(Work{ i, (uint32_t)nq, (uint32_t)ns });
    }
    // the wave-per-pair kernel takes every pair whose subject fits its LDS row buffer; longer subjects go to the lane-per-pair
    // kernel, sorted by size so that the 64 lanes of a wave walk similar matrices; pairs without a cell need no device
    uint32_t wave_ns = ALN_WAVE_NS;
    if (const char *e = getenv("KAAMER_ALIGN_WAVE_NS")) { const long v = atol(e); if (v >= 0 && v <= (long)ALN_WAVE_NS) wave_ns = (uint32_t)v; }
    auto klass = [&](const Work &w) { return (w.nq == 0 || w.ns == 0) ? 2 : (w.ns <= wave_ns ? 0 : 1); };
    std::sort(work.begin(), work.end(), [&](const Work &a, const Work &b) {
        const int ka = klass(a), kb = klass(b);
        if (ka != kb) return ka < kb;
        const uint64_t ca = (uint64_t)a.nq * a.ns, cb = (uint64_t)b.nq * b.ns;
        return ca != cb ? ca > cb : a.pair < b.pair;
    });
    size_t n_wave = 0, n_lane = 0;
    for (const Work &w : work) { if (klass(w) == 0) n_wave++; else if (klass(w) == 1) n_lane++; }
    std::vector<PairDesc> pd(work.size());
    for (size_t i = 0; i < work.size(); i++) {
        const uint32_t q = pair_query[work[i].pair], s = pair_subject[work[i].pair];
        pd[i] = PairDesc{ (uint32_t)offsets[q], work[i].nq, (uint32_t)offsets[s], work[i].ns };
    }
    if (total > 0xFFFFFFFFull) { delete res; return kaamer_fail(KAAMER_E_ARG, "align_pairs: more than 4 GiB of sequence"); }
    std::vector<PairOut> po(work.size());
    memset(po.data(), 0, po.size() * sizeof(PairOut));
    std::vector<std::vector<uint8_t>> ops_of(work.size());
    if (n_wave + n_lane) {
        ALN_HIP(hipSetDevice(device));
        uint64_t budget = 4ull << 30;                  // bytes of direction array per launch
        if (const char *e = getenv("KAAMER_ALIGN_DIR_BYTES")) budget = strtoull(e, nullptr, 10);
        DevBuf<uint8_t> d_codes, d_dirs, d_ops;
        DevBuf<PairDesc> d_pairs;
        DevBuf<WPair> d_wpairs;
        DevBuf<WaveDesc> d_waves;
        DevBuf<int> d_matrix, d_rows;
        DevBuf<PairOut> d_out;
        int rc = d_codes.alloc(codes.size());
        if (!rc) rc = d_pairs.alloc(pd.size());
        if (!rc) rc = d_matrix.alloc(NL * NL);
        if (!rc) rc = d_out.alloc(po.size());
        if (rc) { delete res; return rc; }
        ALN_HIP(hipMemcpy(d_codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice));
        ALN_HIP(hipMemcpy(d_pairs.p, pd.data(), pd.size() * sizeof(PairDesc), hipMemcpyHostToDevice));
        ALN_HIP(hipMemcpy(d_matrix.p, matrix, sizeof matrix, hipMemcpyHostToDevice));
        // ---- wave per pair
        size_t at = 0;
        while (at < n_wave) {
            std::vector<WPair> wp;
            uint64_t dirs = 0, opsb = 0;
            size_t end = at;
            while (end < n_wave) {
                const PairDesc &x = pd[end];
                const uint64_t dbytes = (uint64_t)((x.nq + 63) / 64) * (x.ns + 63) * 64;
                if (!wp.empty() && dirs + dbytes > budget) break;
                wp.push_back(WPair{ x.q_off, x.nq, x.s_off, x.ns, dirs, opsb });
                dirs += dbytes;
                opsb += (uint64_t)x.nq + x.ns;
                end++;
            }
            rc = d_wpairs.alloc(wp.size());
            if (!rc) rc = d_dirs.alloc((size_t)dirs);
            if (!rc) rc = d_ops.alloc((size_t)opsb);
            if (rc) { delete res; return rc; }
            ALN_HIP(hipMemcpy(d_wpairs.p, wp.data(), wp.size() * sizeof(WPair), hipMemcpyHostToDevice));
            WaveParams vp;
            vp.codes = d_codes.p; vp.pairs = d_wpairs.p; vp.matrix = d_matrix.p; vp.gap_open = dp_open;
            vp.dirs = d_dirs.p; vp.ops = d_ops.p; vp.out = d_out.p; vp.first = (uint32_t)at;
            hipLaunchKernelGGL(align_wave_kernel, dim3((unsigned)wp.size()), dim3(64), 0, 0, vp);
            ALN_HIP(hipGetLastError());
            ALN_HIP(hipDeviceSynchronize());
            ALN_HIP(hipMemcpy(po.data() + at, d_out.p + at, (end - at) * sizeof(PairOut), hipMemcpyDeviceToHost));
            std::vector<uint8_t> h_ops((size_t)opsb);
            if (opsb) ALN_HIP(hipMemcpy(h_ops.data(), d_ops.p, (size_t)opsb, hipMemcpyDeviceToHost));
            for (size_t t = at; t < end; t++) {
                const PairOut &o = po[t];
                ops_of[t].assign(h_ops.begin() + (size_t)wp[t - at].ops_off, h_ops.begin() + (size_t)wp[t - at].ops_off + (size_t)o.n_ops);
            }
            at = end;
        }
        // ---- lane per pair (subjects beyond the LDS row buffer)
        const size_t lane_end = n_wave + n_lane;
        while (at < lane_end) {
            // one launch: waves until the direction array is full (a single wave may exceed the budget: it runs alone)
            std::vector<WaveDesc> waves;
            uint64_t rows = 0, dirs = 0, opsb = 0;
            size_t end = at;
            while (end < lane_end) {
                WaveDesc w;
                w.first_pair = (uint32_t)end;
                w.n_pairs = (uint32_t)std::min<size_t>(64, lane_end - end);
                w.max_nq = w.max_ns = 0;
                for (uint32_t t = 0; t < w.n_pairs; t++) { w.max_nq = std::max(w.max_nq, pd[end + t].nq); w.max_ns = std::max(w.max_ns, pd[end + t].ns); }
                const uint64_t dbytes = (uint64_t)w.max_nq * w.max_ns * 64;
                if (!waves.empty() && dirs + dbytes > budget) break;
                w.row_off = rows; w.dir_off = dirs; w.ops_off = opsb;
                rows += 3ull * (w.max_ns + 1) * 64;
                dirs += dbytes;
                opsb += ((uint64_t)w.max_nq + w.max_ns) * 64;
                waves.push_back(w);
                end += w.n_pairs;
            }
            rc = d_waves.alloc(waves.size());
            if (!rc) rc = d_rows.alloc((size_t)rows);
            if (!rc) rc = d_dirs.alloc((size_t)dirs);
            if (!rc) rc = d_ops.alloc((size_t)opsb);
            if (rc) { delete res; return rc; }
            ALN_HIP(hipMemcpy(d_waves.p, waves.data(), waves.size() * sizeof(WaveDesc), hipMemcpyHostToDevice));
            AlignParams ap;
            ap.codes = d_codes.p; ap.pairs = d_pairs.p; ap.waves = d_waves.p; ap.matrix = d_matrix.p; ap.gap_open = dp_open;
            ap.rows = d_rows.p; ap.dirs = d_dirs.p; ap.ops = d_ops.p; ap.out = d_out.p;
            hipLaunchKernelGGL(align_kernel, dim3((unsigned)waves.size()), dim3(64), 0, 0, ap);
            ALN_HIP(hipGetLastError());
            ALN_HIP(hipDeviceSynchronize());
            ALN_HIP(hipMemcpy(po.data() + at, d_out.p + at, (end - at) * sizeof(PairOut), hipMemcpyDeviceToHost));
            std::vector<uint8_t> h_ops((size_t)opsb);
            ALN_HIP(hipMemcpy(h_ops.data(), d_ops.p, (size_t)opsb, hipMemcpyDeviceToHost));
            for (const WaveDesc &w : waves)
                for (uint32_t t = 0; t < w.n_pairs; t++) {
                    const PairOut &o = po[w.first_pair + t];
                    std::vector<uint8_t> &v = ops_of[w.first_pair + t];
                    v.resize((size_t)o.n_ops);
                    for (int x = 0; x < o.n_ops; x++) v[(size_t)x] = h_ops[(size_t)(w.ops_off + (uint64_t)x * 64 + t)];
                }
            at = end;
        }
    }
    // ---- align.Format and align.go:70-160 on the host, from the operations
    for (size_t wi = 0; wi < work.size(); wi++) {
        const uint32_t pi = work[wi].pair;
        kaamer_alignment &a = res->items[pi];
        const PairOut &o = po[wi];
        const std::vector<uint8_t> &ops = ops_of[wi];
        const uint32_t q = pair_query[pi], s = pair_subject[pi];
        const uint8_t *qa = seqs + offsets[q], *sb = seqs + offsets[s];
        auto qchar = [&](int i) { const int c = qa[i]; return (char)((c == 'u' || c == 'U') ? '*' : c); };
        auto schar = [&](int j) { const int c = sb[j]; return (char)((c == 'u' || c == 'U') ? '*' : c); };
        const int len = o.n_ops;
        a.aln_off = res->text.size();
        res->text.resize(res->text.size() + 3 * (size_t)len);
        char *row_a = res->text.data() + a.aln_off, *row_m = row_a + len, *row_b = row_a + 2 * len;
        float identity = 0, similarity = 0, nb_pos = 0;
        int mismatches = 0, ai = o.start_i, bj = o.start_j;
        for (int t = 0; t < len; t++) {
            const uint8_t op = ops[(size_t)(len - 1 - t)];
            const char ca = op == 'L' ? '-' : qchar(ai++), cb = op == 'U' ? '-' : schar(bj++);
            row_a[t] = ca; row_b[t] = cb;
            if (cb == ca) { identity += 1; similarity += 1; row_m[t] = cb; }                 // align.go:87-90
            else {
                if (cb != '-' && ca != '-') mismatches += 1;                                // align.go:92-94
                const int ib = letter_index(cb), ia = letter_index(ca);
                if (matrix[(ib < 0 ? 0 : ib) * NL + (ia < 0 ? 0 : ia)] > 0) { similarity += 1; row_m[t] = '+'; }   // GetAlnScoreAA > 0
                else row_m[t] = ' ';
            }
            nb_pos += 1;
        }
        identity = (identity / nb_pos) * 100;                                               // align.go:101-102 (float32)
        similarity = (similarity / nb_pos) * 100;
        // align.go:105-133 over the feature pairs = maximal runs of one operation
        int raw = 0, gap_openings = 0;
        ai = o.start_i; bj = o.start_j;
        for (int t = 0; t < len;) {
            const uint8_t op = ops[(size_t)(len - 1 - t)];
            int run = 0, score = 0;
            while (t + run < len && ops[(size_t)(len - 1 - t - run)] == op) {
                if (op == 'M') score += matrix[codes[(size_t)(offsets[q] + ai)] * NL + codes[(size_t)(offsets[s] + bj)]];
                else if (op == 'U') score += matrix[codes[(size_t)(offsets[q] + ai)] * NL];
                else score += matrix[codes[(size_t)(offsets[s] + bj)]];
                if (op != 'L') ai++;
                if (op != 'U') bj++;
                run++;
            }
            if (op != 'M') score += dp_open;
            raw += score;
            if (score == -gap_open) {                                                       // align.go:127
                gap_openings += 1;
                raw -= (run - 1) * gap_extend;                                              // align.go:129-130
            }
            t += run;
        }
        a.identity = identity;
        a.similarity = similarity;
        a.length = len;
        a.mismatches = mismatches;
        a.gap_openings = gap_openings;
        a.raw = raw;
        a.bitscore = ((lambda * (double)raw) - std::log(kk)) / std::log(2.0);               // align.go:137
        a.evalue = (double)(offsets[q + 1] - offsets[q]) * (double)number_of_aa / std::pow(2.0, a.bitscore);   // align.go:142
        a.query_start = len ? o.start_i + 1 : 1;                                            // align.go:153-156
        a.query_end = len ? o.end_i : 0;
        a.subject_start = len ? o.start_j + 1 : 1;
        a.subject_end = len ? o.end_j : 0;
        a.status = 0;
    }
    *out = res;
    return KAAMER_OK;
}

}  // extern "C"
