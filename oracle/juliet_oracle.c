/*
 * juliet_oracle.c — CPU restatement of juliet's call+phase hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; nothing under minorseq_amd/ or the C-ABI may link, import or execute it.
 *
 * PARITY UNPINNED.  The reference snapshot (/root/reference) is documentation only: there is no
 * juliet source, binary, test or fixture to compile, import or diff against.
 * Every function below restates the behaviour the reference TEXT documents, citing
 * doc/JULIET.md as J:line, and follows docs/SPEC.md for each constant the text leaves open.
 * What pins it: (i) tests/golden/fisher_golden.json (mpmath 50-digit hypergeometric tails, made by
 * tests/golden/make_fisher_golden.py), (ii) the screenshot relationships of SURVEY.md Appendix A
 * (tests/test_oracle_golden_rows.py: every printed row, the A-I table, the FAQ scenarios).
 *
 * Deliberately simple: by-row uint8 MSA, plain loops, long-double lgammal for the test.
 * Independent of the device code: shares no header with it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* 1 = the plain single-threaded restatement (default).  > 1 = the "all host cores" CPU baseline of SURVEY.md §8d: OpenMP over
 * COLUMNS (pileup) and codon positions (histograms) of a column-major copy of the matrix registered with orc_set_columns —
 * every thread owns its columns' counters, nothing is merged, the sweeps are contiguous streams of host memory — and over
 * reads for the per-read patterns of the phasing stage.  Without a registered copy the sweeps split over reads with
 * per-thread counters summed afterwards.  Results are identical either way (tests/test_oracle_pipeline.py). */
static int g_threads = 1;
static const uint8_t *g_cols = NULL;   /* [n_cols][n_reads]: column c = n_reads consecutive codes */
static uint64_t g_cols_reads = 0;
static uint32_t g_cols_cols = 0;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
/* by_column = NULL forgets the copy.  The caller keeps it alive and in step with the by-row matrix it passes to the calls. */
void orc_set_columns(const uint8_t *by_column, uint64_t n_reads, uint32_t n_cols)
{
    g_cols = by_column;
    g_cols_reads = n_reads;
    g_cols_cols = n_cols;
}
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

enum { SYM_A = 0, SYM_C = 1, SYM_G = 2, SYM_T = 3, SYM_GAP = 4, SYM_MASK = 5, SYM_NONE = 6 };

typedef struct {
    uint32_t begin; /* 1-based, inclusive  (J:134-136) */
    uint32_t end;   /* 1-based, exclusive */
} orc_gene;

typedef struct {
    double match;
    double substitution;
    double deletion; /* reserved: indels are ignored (J:26-27) */
} orc_error_model;

typedef struct {
    double alpha;           /* SPEC §5, default 0.01 */
    double n_tests;         /* Bonferroni factor; <=0 => sum of codons over genes */
    orc_error_model err;    /* SPEC §5 */
    int32_t expected_round; /* 0 ceil (default), 1 floor, 2 nearest */
    int32_t tail;           /* 0 greater (default), 1 two-sided */
} orc_params;

/* same field order/size as the device library's row so tests can share one numpy dtype */
typedef struct {
    uint32_t gene;
    uint32_t codon_pos; /* 1-based AA position in the gene (J:96-97) */
    uint32_t col;       /* window column of the codon's first base */
    uint8_t ref_codon;
    uint8_t codon;
    uint16_t flags;
    uint32_t count;
    uint32_t coverage;
    uint32_t expected;
    uint32_t pad_;
    double p_value; /* Bonferroni-adjusted, <= 1 */
    double log_p;   /* ln of the raw p */
} orc_variant;

/* ------------------------------------------------------------------ pileup (J:99-100) */

/* SPEC §2: per column, counts of A C G T - N; uncovered cells are not counted. */
int orc_pileup(const uint8_t *msa, uint64_t n_reads, uint32_t n_cols, uint32_t *col_counts)
{
    memset(col_counts, 0, (size_t)n_cols * 6 * sizeof(uint32_t));
#ifdef _OPENMP
    if (g_threads > 1 && g_cols && g_cols_reads == n_reads && g_cols_cols == n_cols) {
        /* a thread per range of columns, private counters in registers, one contiguous stream per column */
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int64_t c = 0; c < (int64_t)n_cols; ++c) {
            const uint8_t *col = g_cols + (uint64_t)c * n_reads;
            /* four counter sets taken in turn: nearly every cell of a column holds the same symbol, and one counter
             * incremented cell after cell is a chain of dependent read-modify-writes */
            uint32_t k[4][8];
            memset(k, 0, sizeof k);
            uint64_t i = 0;
            for (; i + 4 <= n_reads; i += 4) {
                k[0][col[i] & 7]++;
                k[1][col[i + 1] & 7]++;
                k[2][col[i + 2] & 7]++;
                k[3][col[i + 3] & 7]++;
            }
            for (; i < n_reads; ++i) k[0][col[i] & 7]++;
            for (int s6 = 0; s6 < 6; ++s6) col_counts[(size_t)c * 6 + s6] = k[0][s6] + k[1][s6] + k[2][s6] + k[3][s6];
        }
        return 0;
    }
    if (g_threads > 1) {
#pragma omp parallel num_threads(g_threads)
        {
            uint32_t *loc = (uint32_t *)calloc((size_t)n_cols * 6, sizeof(uint32_t));
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n_reads; ++i) {
                const uint8_t *row = msa + (uint64_t)i * n_cols;
                for (uint32_t c = 0; c < n_cols; ++c) {
                    uint8_t s = row[c];
                    if (s < 6) loc[(size_t)c * 6 + s]++;
                }
            }
#pragma omp critical
            for (size_t k = 0; k < (size_t)n_cols * 6; ++k) col_counts[k] += loc[k];
            free(loc);
        }
        return 0;
    }
#endif
    for (uint64_t i = 0; i < n_reads; ++i) {
        const uint8_t *row = msa + i * n_cols;
        for (uint32_t c = 0; c < n_cols; ++c) {
            uint8_t s = row[c];
            if (s < 6) col_counts[(size_t)c * 6 + s]++;
        }
    }
    return 0;
}

/* SPEC §3: codon histogram + coverage at given start columns (J:21-27, 94-98, 256-259). */
int orc_codon_hist(const uint8_t *msa, uint64_t n_reads, uint32_t n_cols, const uint32_t *start_cols,
                   uint32_t n_pos, uint32_t *hist, uint32_t *coverage)
{
    memset(hist, 0, (size_t)n_pos * 64 * sizeof(uint32_t));
    memset(coverage, 0, (size_t)n_pos * sizeof(uint32_t));
#ifdef _OPENMP
    if (g_threads > 1 && g_cols && g_cols_reads == n_reads && g_cols_cols == n_cols) {
        /* a thread per range of positions: three contiguous column streams, the position's own 64 bins */
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int64_t p = 0; p < (int64_t)n_pos; ++p) {
            const uint32_t c = start_cols[p];
            if ((uint64_t)c + 2 >= n_cols) continue;
            const uint8_t *c0 = g_cols + (uint64_t)c * n_reads, *c1 = c0 + n_reads, *c2 = c1 + n_reads;
            /* bin 64 takes the reads that are not in the coverage (gap, N or uncovered); four histograms in turn, as above */
            uint32_t h[4][65], cov = 0;
            memset(h, 0, sizeof h);
            uint64_t i = 0;
#define ORC_BIN(j) (((c0[j] | c1[j] | c2[j]) > 3) ? 64u : (16u * c0[j] + 4u * c1[j] + c2[j]))
            for (; i + 4 <= n_reads; i += 4) {
                h[0][ORC_BIN(i)]++;
                h[1][ORC_BIN(i + 1)]++;
                h[2][ORC_BIN(i + 2)]++;
                h[3][ORC_BIN(i + 3)]++;
            }
            for (; i < n_reads; ++i) h[0][ORC_BIN(i)]++;
#undef ORC_BIN
            for (int j = 0; j < 64; ++j) {
                const uint32_t v = h[0][j] + h[1][j] + h[2][j] + h[3][j];
                hist[(size_t)p * 64 + j] = v;
                cov += v;
            }
            coverage[p] = cov;
        }
        return 0;
    }
    if (g_threads > 1) {
#pragma omp parallel num_threads(g_threads)
        {
            uint32_t *lh = (uint32_t *)calloc((size_t)n_pos * 64 + 1, sizeof(uint32_t));
            uint32_t *lc = (uint32_t *)calloc((size_t)n_pos + 1, sizeof(uint32_t));
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n_reads; ++i) {
                const uint8_t *row = msa + (uint64_t)i * n_cols;
                for (uint32_t p = 0; p < n_pos; ++p) {
                    uint32_t c = start_cols[p];
                    if ((uint64_t)c + 2 >= n_cols) continue;
                    uint8_t s0 = row[c], s1 = row[c + 1], s2 = row[c + 2];
                    if (s0 > 3 || s1 > 3 || s2 > 3) continue;
                    lh[(size_t)p * 64 + 16 * s0 + 4 * s1 + s2]++;
                    lc[p]++;
                }
            }
#pragma omp critical
            {
                for (size_t k = 0; k < (size_t)n_pos * 64; ++k) hist[k] += lh[k];
                for (size_t k = 0; k < n_pos; ++k) coverage[k] += lc[k];
            }
            free(lh);
            free(lc);
        }
        return 0;
    }
#endif
    /* reads outermost: one sequential sweep of the by-row matrix */
    for (uint64_t i = 0; i < n_reads; ++i) {
        const uint8_t *row = msa + i * n_cols;
        for (uint32_t p = 0; p < n_pos; ++p) {
            uint32_t c = start_cols[p];
            if ((uint64_t)c + 2 >= n_cols) continue;
            uint8_t s0 = row[c], s1 = row[c + 1], s2 = row[c + 2];
            if (s0 > 3 || s1 > 3 || s2 > 3) continue; /* gap, N or uncovered: not in coverage */
            hist[(size_t)p * 64 + 16 * s0 + 4 * s1 + s2]++;
            coverage[p]++;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ Fisher's exact (J:38-42) */

static long double lfactl(uint64_t n) { return lgammal((long double)n + 1.0L); }

/* ln P(X = x) for X ~ Hypergeometric(M, K, n) */
static long double hyper_logpmf(uint64_t x, uint64_t M, uint64_t K, uint64_t n)
{
    return lfactl(K) + lfactl(M - K) + lfactl(n) + lfactl(M - n) - lfactl(M) - lfactl(x) - lfactl(K - x) -
           lfactl(n - x) - lfactl(M - K - n + x);
}

/* Upper tail P(X >= a) of the 2x2 table [[a,b],[c,d]]; also returns ln p.  One-sided "greater". */
static long double fisher_greater_l(uint64_t a, uint64_t b, uint64_t c, uint64_t d, long double *logp)
{
    uint64_t M = a + b + c + d, K = a + c, n = a + b;
    uint64_t hi = K < n ? K : n;
    uint64_t lo = (K + n > M) ? (K + n - M) : 0;
    if (a <= lo) { *logp = 0.0L; return 1.0L; }
    /* a above the mean K*n/M: sum the (decreasing) upper tail directly; else 1 - lower tail */
    long double mean = (long double)K * (long double)n / (long double)M;
    if ((long double)a > mean) {
        long double l0 = hyper_logpmf(a, M, K, n);
        long double term = 1.0L, sum = 1.0L;
        for (uint64_t x = a; x < hi; ++x) {
            term *= ((long double)(K - x) * (long double)(n - x)) /
                    ((long double)(x + 1) * (long double)(M - K - n + x + 1));
            sum += term;
            if (term < sum * 1e-25L) break;
        }
        *logp = l0 + logl(sum);
        return expl(*logp);
    } else {
        /* lower tail P(X <= a-1), summed downwards from a-1 (terms decrease below the mode) */
        uint64_t x0 = a - 1;
        long double l0 = hyper_logpmf(x0, M, K, n);
        long double term = 1.0L, sum = 1.0L;
        for (uint64_t x = x0; x > lo; --x) {
            term *= ((long double)x * (long double)(M - K - n + x)) /
                    ((long double)(K - x + 1) * (long double)(n - x + 1));
            sum += term;
            if (term < sum * 1e-25L) break;
        }
        long double lower = expl(l0 + logl(sum));
        long double p = 1.0L - lower;
        if (p < 0.0L) p = 0.0L;
        *logp = log1pl(-lower);
        return p;
    }
}

/* two-sided: sum of all table probabilities <= P(observed) (the usual definition), brute force */
static long double fisher_two_sided_l(uint64_t a, uint64_t b, uint64_t c, uint64_t d, long double *logp)
{
    uint64_t M = a + b + c + d, K = a + c, n = a + b;
    uint64_t hi = K < n ? K : n;
    uint64_t lo = (K + n > M) ? (K + n - M) : 0;
    long double lobs = hyper_logpmf(a, M, K, n);
    long double sum = 0.0L;
    for (uint64_t x = lo; x <= hi; ++x) {
        long double l = hyper_logpmf(x, M, K, n);
        if (l <= lobs + 1e-7L * fabsl(lobs) + 1e-30L) sum += expl(l - lobs);
    }
    *logp = lobs + logl(sum);
    if (*logp > 0.0L) *logp = 0.0L;
    return expl(*logp);
}

double orc_fisher(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int32_t tail, double *log_p)
{
    long double lp;
    long double p = tail == 1 ? fisher_two_sided_l(a, b, c, d, &lp) : fisher_greater_l(a, b, c, d, &lp);
    if (log_p) *log_p = (double)lp;
    if (p > 1.0L) p = 1.0L;
    return (double)p;
}

/* SPEC §5: expected count under the error model */
static double codon_error_prob(const orc_error_model *em, int ref, int j)
{
    double p = 1.0;
    for (int i = 0; i < 3; ++i) {
        int shift = 4 - 2 * i;
        int rb = (ref >> shift) & 3, jb = (j >> shift) & 3;
        p = p * (rb == jb ? em->match : em->substitution);
    }
    return p;
}

uint32_t orc_expected(const orc_params *prm, uint32_t coverage, int ref, int j)
{
    double x = (double)coverage * codon_error_prob(&prm->err, ref, j);
    double r = prm->expected_round == 1 ? floor(x) : prm->expected_round == 2 ? floor(x + 0.5) : ceil(x);
    if (r < 0.0) r = 0.0;
    if (r > (double)coverage) r = (double)coverage;
    return (uint32_t)r;
}

double orc_default_n_tests(const orc_gene *genes, uint32_t n_genes)
{
    double n = 0.0;
    for (uint32_t g = 0; g < n_genes; ++g)
        if (genes[g].end > genes[g].begin) n += (double)((genes[g].end - genes[g].begin) / 3);
    return n;
}

/* ------------------------------------------------------------------ call (J:38-42, 94-98, 133-134) */

/*
 * refseq: base codes (0..3, anything else = non-ACGT) of the WHOLE reference, 0-based, or NULL for
 * majority-codon mode.  Returns 0, or 1 if the table would overflow `cap` (n_out = rows needed).
 */
int orc_call(const uint8_t *msa, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, const orc_gene *genes,
             uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len, const orc_params *prm, orc_variant *out,
             uint32_t cap, uint32_t *n_out)
{
    double n_tests = prm->n_tests > 0.0 ? prm->n_tests : orc_default_n_tests(genes, n_genes);
    /* 1. evaluated positions (SPEC §3) */
    size_t max_pos = 0;
    for (uint32_t g = 0; g < n_genes; ++g)
        if (genes[g].end > genes[g].begin && genes[g].begin > 0) max_pos += (genes[g].end - genes[g].begin) / 3;
    uint32_t *pg = (uint32_t *)malloc((max_pos + 1) * sizeof(uint32_t));
    uint32_t *pk = (uint32_t *)malloc((max_pos + 1) * sizeof(uint32_t));
    uint32_t *pc = (uint32_t *)malloc((max_pos + 1) * sizeof(uint32_t));
    uint32_t P = 0;
    for (uint32_t g = 0; g < n_genes; ++g) {
        if (genes[g].end <= genes[g].begin || genes[g].begin == 0) continue;
        uint32_t ncod = (genes[g].end - genes[g].begin) / 3;
        for (uint32_t k = 0; k < ncod; ++k) {
            int64_t r = (int64_t)genes[g].begin - 1 + 3 * (int64_t)k; /* 0-based reference coordinate */
            int64_t c = r - (int64_t)win_begin;
            if (c < 0 || c + 2 >= (int64_t)n_cols) continue;
            pg[P] = g; pk[P] = k; pc[P] = (uint32_t)c; ++P;
        }
    }
    /* 2. one sweep over the reads for all histograms */
    uint32_t *hists = (uint32_t *)malloc(((size_t)P + 1) * 64 * sizeof(uint32_t));
    uint32_t *covs = (uint32_t *)malloc(((size_t)P + 1) * sizeof(uint32_t));
    orc_codon_hist(msa, n_reads, n_cols, pc, P, hists, covs);
    /* 3. reference codon, test, rows in (gene, k, codon) order */
    uint32_t n = 0;
    for (uint32_t q = 0; q < P; ++q) {
        const uint32_t *hist = hists + (size_t)q * 64;
        uint32_t cov = covs[q], g = pg[q], k = pk[q], cc = pc[q];
        int64_t r = (int64_t)cc + win_begin;
        int ref;
        if (refseq) {
            if ((uint64_t)r + 2 >= ref_len) continue;
            if (refseq[r] > 3 || refseq[r + 1] > 3 || refseq[r + 2] > 3) continue;
            ref = 16 * refseq[r] + 4 * refseq[r + 1] + refseq[r + 2];
        } else {
            if (cov == 0) continue;
            ref = 0;
            for (int j = 1; j < 64; ++j)
                if (hist[j] > hist[ref]) ref = j;
        }
        for (int j = 0; j < 64; ++j) {
            if (j == ref || hist[j] == 0) continue;
            uint32_t e = orc_expected(prm, cov, ref, j);
            double lp;
            double p = orc_fisher(hist[j], cov - hist[j], e, cov - e, prm->tail, &lp);
            double padj = p * n_tests;
            if (padj > 1.0) padj = 1.0;
            if (!(padj < prm->alpha)) continue;
            if (n < cap) {
                orc_variant *v = &out[n];
                memset(v, 0, sizeof(*v));
                v->gene = g;
                v->codon_pos = k + 1;
                v->col = cc;
                v->ref_codon = (uint8_t)ref;
                v->codon = (uint8_t)j;
                v->count = hist[j];
                v->coverage = cov;
                v->expected = e;
                v->p_value = padj;
                v->log_p = lp;
            }
            ++n;
        }
    }
    free(pg); free(pk); free(pc); free(hists); free(covs);
    *n_out = n;
    return n > cap ? 1 : 0;
}

/* ------------------------------------------------------------------ phasing (J:192-211, 253-254, 372-381) */

enum { FLAG_GAP = 1, FLAG_HET = 2, FLAG_PARTIAL = 4 };
enum { ORC_MAX_HAP = 702, HAP_INSUFFICIENT = 0xFFFE, HAP_DAMAGED = 0xFFFF };

typedef struct {
    uint32_t reported_reads;
    uint32_t insufficient_reads;
    uint32_t damaged_reads;
    uint32_t marginal_gap;
    uint32_t marginal_heteroduplex;
    uint32_t marginal_partial;
    uint32_t n_positions; /* Vp */
    uint32_t n_haplotypes;
} orc_phase_summary;

static uint32_t g_vp;              /* pattern width for the qsort comparator */
static const uint8_t *g_patterns;  /* [n_clean][vp] */

static int cmp_read_pattern(const void *x, const void *y)
{
    uint32_t i = *(const uint32_t *)x, j = *(const uint32_t *)y;
    int r = memcmp(g_patterns + (size_t)i * g_vp, g_patterns + (size_t)j * g_vp, g_vp);
    if (r) return r;
    return i < j ? -1 : i > j;
}

typedef struct {
    uint32_t count, first_read, first_slot, id;
} grp;

/* SPEC §8 order: count descending, then pattern ascending (codon indices compared position by position) */
static int cmp_grp(const void *x, const void *y)
{
    const grp *a = (const grp *)x, *b = (const grp *)y;
    if (a->count != b->count) return a->count > b->count ? -1 : 1;
    return memcmp(g_patterns + (size_t)a->first_read * g_vp, g_patterns + (size_t)b->first_read * g_vp, g_vp);
}

/*
 * variants: filtered table (only .col and .codon are read).  Outputs (caller-allocated):
 *   pos_cols[n_var]           distinct variant columns, ascending (first Vp entries valid)
 *   hap_count[702], hap_first[702], hap_pattern[702][Vp'] with row stride n_var
 *   hit[n_var][702] row stride ORC_MAX_HAP, read_hap[n_reads], cooc[n_var][n_var]
 */
int orc_phase(const uint8_t *msa, uint64_t n_reads, uint32_t n_cols, const orc_variant *variants, uint32_t n_var,
              uint32_t min_reads, orc_phase_summary *sum, uint32_t *pos_cols, uint32_t *hap_count,
              uint32_t *hap_first, uint8_t *hap_pattern, uint8_t *hit, uint16_t *read_hap, uint32_t *cooc)
{
    memset(sum, 0, sizeof(*sum));
    /* distinct columns, ascending */
    uint32_t vp = 0;
    for (uint32_t v = 0; v < n_var; ++v) {
        uint32_t c = variants[v].col, k = 0;
        while (k < vp && pos_cols[k] != c) ++k;
        if (k == vp) pos_cols[vp++] = c;
    }
    for (uint32_t i = 1; i < vp; ++i) { /* insertion sort */
        uint32_t c = pos_cols[i], j = i;
        while (j > 0 && pos_cols[j - 1] > c) { pos_cols[j] = pos_cols[j - 1]; --j; }
        pos_cols[j] = c;
    }
    sum->n_positions = vp;
    if (cooc) memset(cooc, 0, (size_t)n_var * n_var * sizeof(uint32_t));
    if (hit) memset(hit, 0, (size_t)n_var * ORC_MAX_HAP);
    if (vp == 0) {
        for (uint64_t i = 0; i < n_reads; ++i) read_hap[i] = HAP_DAMAGED;
        return 0;
    }

    uint8_t *patterns = (uint8_t *)malloc((size_t)n_reads * vp);
    uint32_t *clean = (uint32_t *)malloc((size_t)n_reads * sizeof(uint32_t));
    uint32_t n_clean = 0;
    /* per read: pattern + damage flags (independent of each other: threads split the reads when there are several);
     * the categories and the list of clean reads follow in read order */
    uint8_t *rflags = (uint8_t *)malloc((size_t)n_reads + 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int64_t i = 0; i < (int64_t)n_reads; ++i) {
        const uint8_t *row = msa + (uint64_t)i * n_cols;
        unsigned flags = 0;
        uint8_t *pat = patterns + (size_t)i * vp;
        for (uint32_t p = 0; p < vp; ++p) {
            uint32_t c = pos_cols[p];
            uint8_t s[3];
            for (int t = 0; t < 3; ++t) {
                s[t] = (c + t < n_cols) ? row[c + t] : SYM_NONE;
                if (s[t] == SYM_NONE) flags |= FLAG_PARTIAL;
                else if (s[t] == SYM_GAP) flags |= FLAG_GAP;
                else if (s[t] == SYM_MASK) flags |= FLAG_HET;
            }
            pat[p] = (uint8_t)(16 * (s[0] & 3) + 4 * (s[1] & 3) + (s[2] & 3));
        }
        rflags[i] = (uint8_t)flags;
    }
    for (uint64_t i = 0; i < n_reads; ++i) {
        const unsigned flags = rflags[i];
        if (flags) {
            sum->damaged_reads++;
            if (flags & FLAG_GAP) sum->marginal_gap++;
            if (flags & FLAG_HET) sum->marginal_heteroduplex++;
            if (flags & FLAG_PARTIAL) sum->marginal_partial++;
            read_hap[i] = HAP_DAMAGED;
        } else {
            clean[n_clean++] = (uint32_t)i;
            read_hap[i] = HAP_INSUFFICIENT;
        }
    }
    free(rflags);

    grp *groups = (grp *)malloc(((size_t)n_clean + 1) * sizeof(grp));
    uint32_t n_groups = 0;
    uint32_t *gid_of = NULL;   /* all-cores form: group of every clean read */
    g_vp = vp;
    g_patterns = patterns;
    if (g_threads > 1) {
        /* The all-cores baseline groups in one pass over the clean reads, in read order: an open-addressing table keyed
         * by the pattern's bytes whose slots hold a group number; equality is decided by comparing whole patterns (never
         * by the hash), the first read that brings a pattern is the group's first read.  Same groups, same counts, same
         * first reads as the sort below — which costs n log n pattern comparisons on one thread. */
        uint64_t M = 64;
        while (M < 2 * (uint64_t)n_clean) M <<= 1;
        uint32_t *slot = (uint32_t *)malloc(M * sizeof(uint32_t));
        memset(slot, 0xFF, M * sizeof(uint32_t));
        gid_of = (uint32_t *)malloc(((size_t)n_clean + 1) * sizeof(uint32_t));
        for (uint32_t s = 0; s < n_clean; ++s) {
            const uint8_t *pat = patterns + (size_t)clean[s] * vp;
            uint64_t h = 1469598103934665603ull;
            for (uint32_t p = 0; p < vp; ++p) h = (h ^ pat[p]) * 1099511628211ull;
            uint64_t at = (h ^ (h >> 29)) & (M - 1);
            for (;;) {
                const uint32_t g = slot[at];
                if (g == 0xFFFFFFFFu) {
                    slot[at] = n_groups;
                    groups[n_groups].count = 1;
                    groups[n_groups].first_read = clean[s];
                    groups[n_groups].first_slot = s;
                    groups[n_groups].id = n_groups;
                    gid_of[s] = n_groups++;
                    break;
                }
                if (memcmp(patterns + (size_t)groups[g].first_read * vp, pat, vp) == 0) {
                    groups[g].count++;
                    gid_of[s] = g;
                    break;
                }
                at = (at + 1) & (M - 1);
            }
        }
        free(slot);
    } else {
        /* exact grouping: sort clean reads by (pattern, index), run-length */
        qsort(clean, n_clean, sizeof(uint32_t), cmp_read_pattern);
        for (uint32_t s = 0; s < n_clean;) {
            uint32_t e = s + 1;
            while (e < n_clean &&
                   memcmp(patterns + (size_t)clean[s] * vp, patterns + (size_t)clean[e] * vp, vp) == 0)
                ++e;
            groups[n_groups].count = e - s;
            groups[n_groups].first_read = clean[s]; /* smallest index: sort is by (pattern, index) */
            groups[n_groups].first_slot = s;
            groups[n_groups].id = n_groups;
            ++n_groups;
            s = e;
        }
    }
    qsort(groups, n_groups, sizeof(grp), cmp_grp);

    uint32_t H = 0;
    uint16_t *hap_of_gid = gid_of ? (uint16_t *)malloc(((size_t)n_groups + 1) * sizeof(uint16_t)) : NULL;
    for (uint32_t gi = 0; gi < n_groups; ++gi) {
        const grp *G = &groups[gi];
        if (G->count >= min_reads && H < ORC_MAX_HAP) {
            hap_count[H] = G->count;
            hap_first[H] = G->first_read;
            memcpy(hap_pattern + (size_t)H * n_var, patterns + (size_t)G->first_read * vp, vp);
            if (hap_of_gid) hap_of_gid[G->id] = (uint16_t)H;
            else
                for (uint32_t s = G->first_slot; s < G->first_slot + G->count; ++s) read_hap[clean[s]] = (uint16_t)H;
            sum->reported_reads += G->count;
            ++H;
        } else {
            if (hap_of_gid) hap_of_gid[G->id] = HAP_INSUFFICIENT;
            sum->insufficient_reads += G->count;
        }
    }
    if (gid_of) {
        for (uint32_t s = 0; s < n_clean; ++s) read_hap[clean[s]] = hap_of_gid[gid_of[s]];
        free(hap_of_gid);
        free(gid_of);
    }
    sum->n_haplotypes = H;

    for (uint32_t v = 0; v < n_var; ++v) {
        uint32_t p = 0;
        while (pos_cols[p] != variants[v].col) ++p;
        for (uint32_t h = 0; h < H; ++h)
            hit[(size_t)v * ORC_MAX_HAP + h] = hap_pattern[(size_t)h * n_var + p] == variants[v].codon;
    }
    if (cooc)
        for (uint32_t v = 0; v < n_var; ++v)
            for (uint32_t w = 0; w < n_var; ++w) {
                uint32_t s = 0;
                for (uint32_t h = 0; h < H; ++h)
                    if (hit[(size_t)v * ORC_MAX_HAP + h] && hit[(size_t)w * ORC_MAX_HAP + h]) s += hap_count[h];
                cooc[(size_t)v * n_var + w] = s;
            }
    free(groups);
    free(clean);
    free(patterns);
    return 0;
}

/* sizes, so the ctypes side can assert its dtype matches */
uint32_t orc_sizeof_variant(void) { return (uint32_t)sizeof(orc_variant); }
uint32_t orc_sizeof_params(void) { return (uint32_t)sizeof(orc_params); }

/* ------------------------------------------------------------------ fuse-style consensus (doc/FUSE.md:17-20) */

/*
 * "Fuse includes in-frame insertions with a certain distance to each other.  Major deletions are being removed."
 * Insertions are not part of the MSA (J:26-27), so they are counted from the aligned records: an insertion (cigar I)
 * sits BEFORE the window column of the next reference base.  len_hist[c][min(len, 31)] counts insertions by length,
 * base_counts[c][j][b] the inserted bases at offset j < 30 (b = 0..3; other letters are not counted).
 * pos: 0-based leftmost reference position; cigar words (len << 4 | op); seq4: BAM's packed bases, read r from byte
 * seq_off[r], first base in the high nibble.  SPEC §11.
 */
enum { ORC_INS_LEN_BINS = 32, ORC_INS_MAX_BASES = 30 };

int orc_insertions(uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, const int32_t *pos, const uint32_t *cigar,
                   const uint64_t *cig_off, const uint8_t *seq4, const uint64_t *seq_off, uint32_t *len_hist,
                   uint32_t *base_counts)
{
    static const uint8_t code_of[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4}; /* =ACMGRSVTWYHKDBN */
    memset(len_hist, 0, (size_t)n_cols * ORC_INS_LEN_BINS * sizeof(uint32_t));
    memset(base_counts, 0, (size_t)n_cols * ORC_INS_MAX_BASES * 4 * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_reads; ++r) {
        int64_t rp = pos[r];
        uint64_t qp = 0;
        const uint8_t *sq = seq4 + seq_off[r];
        for (uint64_t k = cig_off[r]; k < cig_off[r + 1]; ++k) {
            uint32_t op = cigar[k] & 15u, len = cigar[k] >> 4;
            if (op == 1) { /* I */
                int64_t c = rp - (int64_t)win_begin;
                if (c >= 0 && c < (int64_t)n_cols) {
                    len_hist[(size_t)c * ORC_INS_LEN_BINS + (len < 31 ? len : 31)]++;
                    /* only insertions that can enter a consensus vote on its bases: in-frame, at most 30 long (SPEC 11) */
                    for (uint32_t j = 0; len % 3 == 0 && len <= ORC_INS_MAX_BASES && j < len; ++j) {
                        uint64_t q = qp + j;
                        uint8_t b16 = (q & 1) ? (sq[q >> 1] & 15) : (sq[q >> 1] >> 4);
                        uint8_t b = code_of[b16];
                        if (b < 4) base_counts[((size_t)c * ORC_INS_MAX_BASES + j) * 4 + b]++;
                    }
                }
                qp += len;
            } else if (op == 4) { /* S */
                qp += len;
            } else if (op == 7 || op == 8 || op == 0) { /* = X (M) */
                qp += len;
                rp += len;
            } else if (op == 2 || op == 3) { /* D N */
                rp += len;
            }
        }
    }
    return 0;
}

/*
 * The consensus of a window: per column the majority of A C G T - (lowest code on ties); a column whose majority is
 * '-' is removed ("major deletions are being removed"), a column nobody covers is N.  Before column c an insertion is
 * included iff an IN-FRAME length L (3, 6, ... 30; the most frequent, the shorter on ties) is carried by more than
 * `min_frac` of the reads covering c and the last included insertion lies at least `min_distance` columns back
 * ("in-frame insertions with a certain distance to each other"; both UNPINNED, SPEC §11); its bases are the majority
 * base per offset.  out: characters, at most n_cols * 31; returns the length.
 */
uint32_t orc_fuse(uint32_t n_cols, const uint32_t *col_counts, const uint32_t *len_hist, const uint32_t *base_counts,
                  double min_frac, uint32_t min_distance, char *out)
{
    uint32_t n = 0;
    int64_t last_ins = -(int64_t)min_distance - 1;
    for (uint32_t c = 0; c < n_cols; ++c) {
        const uint32_t *k = col_counts + (size_t)c * 6;
        uint32_t covering = k[0] + k[1] + k[2] + k[3] + k[4] + k[5];
        if (len_hist && covering) {
            uint32_t bestL = 0, bestN = 0;
            for (uint32_t L = 3; L <= ORC_INS_MAX_BASES; L += 3) {
                uint32_t v = len_hist[(size_t)c * ORC_INS_LEN_BINS + L];
                if (v > bestN) { bestN = v; bestL = L; }
            }
            if (bestL && (double)bestN > min_frac * (double)covering && (int64_t)c - last_ins >= (int64_t)min_distance) {
                for (uint32_t j = 0; j < bestL; ++j) {
                    const uint32_t *b = base_counts + ((size_t)c * ORC_INS_MAX_BASES + j) * 4;
                    uint32_t best = 0;
                    for (uint32_t s = 1; s < 4; ++s)
                        if (b[s] > b[best]) best = s;
                    out[n++] = "ACGT"[best];
                }
                last_ins = c;
            }
        }
        uint32_t best = 0, bv = k[0];
        for (uint32_t s = 1; s < 5; ++s)
            if (k[s] > bv) { bv = k[s]; best = s; }
        if (bv == 0) out[n++] = 'N';
        else if (best < 4) out[n++] = "ACGT"[best];
    }
    return n;
}

/*
 * The same consensus computed a second way, straight from the aligned records and the by-row matrix — no counters: the
 * insertions of every read are collected as explicit records (column, length, bases), sorted by column, and each column is
 * decided from its own records and from a sweep over the matrix rows.  This is the checker of the front end's consensus
 * (minorseq_amd/host/fuse.hpp works from the device's counters; the two share nothing but the rule of SPEC 11).
 * rows: uint8[n_reads][n_cols] symbol codes of the window.  Returns the length written to `out` (at most n_cols * 31).
 */
typedef struct { uint32_t col, len; uint8_t base[ORC_INS_MAX_BASES]; } orc_ins_rec;

static int orc_ins_cmp(const void *a, const void *b)
{
    const orc_ins_rec *x = (const orc_ins_rec *)a, *y = (const orc_ins_rec *)b;
    return x->col < y->col ? -1 : x->col > y->col;
}

uint32_t orc_fuse_records(const uint8_t *rows, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, const int32_t *pos,
                          const uint32_t *cigar, const uint64_t *cig_off, const uint8_t *seq4, const uint64_t *seq_off,
                          double min_frac, uint32_t min_distance, char *out)
{
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    /* pass 1: one record per insertion that starts inside the window */
    size_t cap = 1024, n_rec = 0;
    orc_ins_rec *rec = (orc_ins_rec *)malloc(cap * sizeof *rec);
    for (uint64_t r = 0; r < n_reads && rec; ++r) {
        int64_t ref = pos[r];
        uint64_t q = 0;
        for (uint64_t k = cig_off[r]; k < cig_off[r + 1]; ++k) {
            const uint32_t op = cigar[k] & 15u, len = cigar[k] >> 4;
            const int consumes_query = op == 0 || op == 1 || op == 4 || op == 7 || op == 8;
            const int consumes_ref = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
            if (op == 1 && ref >= (int64_t)win_begin && ref < (int64_t)win_begin + n_cols) {
                if (n_rec == cap) {
                    cap *= 2;
                    rec = (orc_ins_rec *)realloc(rec, cap * sizeof *rec);
                    if (!rec) break;
                }
                orc_ins_rec *x = &rec[n_rec++];
                x->col = (uint32_t)(ref - (int64_t)win_begin);
                x->len = len;
                for (uint32_t j = 0; j < len && j < ORC_INS_MAX_BASES; ++j) {
                    const uint64_t b = q + j;
                    const uint8_t code = (b & 1) ? (uint8_t)(seq4[seq_off[r] + (b >> 1)] & 15) : (uint8_t)(seq4[seq_off[r] + (b >> 1)] >> 4);
                    x->base[j] = (uint8_t)nt16[code];
                }
            }
            if (consumes_query) q += len;
            if (consumes_ref) ref += len;
        }
    }
    if (!rec) return 0;
    qsort(rec, n_rec, sizeof *rec, orc_ins_cmp);
    /* pass 2: column by column */
    uint32_t n = 0;
    size_t at = 0;
    int have_last = 0;
    uint32_t last_col = 0;
    for (uint32_t c = 0; c < n_cols; ++c) {
        uint32_t sym[7] = {0, 0, 0, 0, 0, 0, 0};
        for (uint64_t r = 0; r < n_reads; ++r) sym[rows[r * n_cols + c] < 7 ? rows[r * n_cols + c] : 6]++;
        const uint32_t covering = sym[0] + sym[1] + sym[2] + sym[3] + sym[4] + sym[5];
        const size_t first = at;
        while (at < n_rec && rec[at].col == c) ++at;
        /* the most frequent in-frame length among this column's records (the shorter one on ties) */
        uint32_t bestL = 0, bestN = 0;
        for (uint32_t L = 3; L <= ORC_INS_MAX_BASES; L += 3) {
            uint32_t cnt = 0;
            for (size_t i = first; i < at; ++i) cnt += rec[i].len == L;
            if (cnt > bestN) { bestN = cnt; bestL = L; }
        }
        const int far_enough = !have_last || c - last_col >= min_distance;
        if (covering && bestL && (double)bestN > min_frac * (double)covering && far_enough) {
            for (uint32_t j = 0; j < bestL; ++j) {   /* majority base per offset over the column's in-frame insertions */
                uint32_t votes[4] = {0, 0, 0, 0};
                for (size_t i = first; i < at; ++i) {
                    if (rec[i].len % 3 != 0 || rec[i].len > ORC_INS_MAX_BASES || j >= rec[i].len) continue;
                    const char ch = (char)rec[i].base[j];
                    const int b = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
                    if (b >= 0) votes[b]++;
                }
                int best = 0;
                for (int b = 1; b < 4; ++b)
                    if (votes[b] > votes[best]) best = b;
                out[n++] = "ACGT"[best];
            }
            have_last = 1;
            last_col = c;
        }
        /* the column itself: majority of A C G T - (lowest code on ties); '-' drops it; nobody covering prints N */
        int best = -1;
        uint32_t most = 0;
        for (int b = 0; b < 5; ++b)
            if (sym[b] > most) { most = sym[b]; best = b; }
        if (best < 0) out[n++] = 'N';
        else if (best < 4) out[n++] = "ACGT"[best];
    }
    free(rec);
    return n;
}
