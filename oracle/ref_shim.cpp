/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" doors onto the *compiled reference* (enzorucci/OSWALD host
 * sources built in place from /root/reference by oracle/Makefile into
 * oracle/_ref/).  This file contains no reference code: it only declares
 * the reference's own function prototypes and calls them, so that
 * oracle/gen_golden.py and the oracle-validation tests can obtain the
 * reference's real outputs through ctypes.  It exists only in the build
 * container; the GPU box never has /root/reference and never needs this.
 *
 * Prototypes mirrored (they are compiled as C++ by the recipe, hence no
 * extern "C" on them):
 *   sw_host .......................... host/src/FPGAsearch.h:33
 *   sort_scores ...................... host/src/utils.h:41
 *   preprocess_db .................... host/src/sequences.h:22
 *   load_query_sequences ............. host/src/sequences.h:53
 *   assemble_multiple_chunks_db ...... host/src/sequences.h:25
 *   load_database_headers ............ host/src/sequences.h:50
 *   globals cpu_block_size, blosum62.. host/src/arguments.h:35-40
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void sw_host(char *a, unsigned short int m, char *b, unsigned short int n, char *submat, int *scores, int *overflow,
             char open_gap, char extend_gap, char *scoreProfile, __m128i *row, __m128i *maxRow, __m128i *maxCol, __m128i *lastCol);
void sort_scores(int *scores, char **titles, unsigned long int size, int threads);
void preprocess_db(char *input_filename, char *out_filename, int n_procs);
void load_query_sequences(char *queries_filename, char **ptr_query_sequences, char ***ptr_query_headers,
                          unsigned short int **ptr_query_sequences_lengths, unsigned long int *query_sequences_count,
                          unsigned long int *ptr_Q, unsigned int **ptr_query_sequences_disp, int n_procs);
void assemble_multiple_chunks_db(char *sequences_filename, int vector_length, unsigned long int max_buffer_size, unsigned int num_devices,
                                 unsigned long int *sequences_count, unsigned long int *D, unsigned short int *sequences_db_max_length,
                                 int *max_title_length, unsigned long int *vect_sequences_count, unsigned long int *vD,
                                 char ***ptr_chunk_vect_sequences_db, unsigned int *chunk_count, unsigned int **ptr_chunk_vect_sequences_db_count,
                                 unsigned long int **ptr_chunk_vect_accum_sequences_db_count, unsigned long int **ptr_chunk_vD,
                                 unsigned long int *max_chunk_vD, unsigned short int ***ptr_chunk_vect_sequences_db_lengths,
                                 unsigned short int ***ptr_chunk_nbbs, unsigned int ***ptr_chunk_vect_sequences_db_disp, int n_procs);
void load_database_headers(char *sequences_filename, unsigned long int sequences_count, int max_title_length, char ***ptr_sequences_db_headers);

extern int cpu_block_size;
extern char blosum45[], blosum50[], blosum62[], blosum80[], blosum90[], pam30[], pam70[], pam250[];

extern "C" {

/* 768 bytes of one of the reference's matrices; name as on its command line. */
const char *ref_submat(const char *name)
{
    if (!strcmp(name, "blosum45")) return blosum45;
    if (!strcmp(name, "blosum50")) return blosum50;
    if (!strcmp(name, "blosum62")) return blosum62;
    if (!strcmp(name, "blosum80")) return blosum80;
    if (!strcmp(name, "blosum90")) return blosum90;
    if (!strcmp(name, "pam30")) return pam30;
    if (!strcmp(name, "pam70")) return pam70;
    if (!strcmp(name, "pam250")) return pam250;
    return 0;
}

/* The reference's own int32 SSE4.1 recompute on one 16-lane interleaved
 * group, all four quarters forced (overflow = {1,1,1,1}). */
void ref_sw_host_group(const uint8_t *a, unsigned short m, const uint8_t *b16, unsigned short n, const char *submat,
                       int open_gap, int extend_gap, int block, int *scores16)
{
    cpu_block_size = block;
    int overflow[4] = {1, 1, 1, 1};
    char *sp;
    __m128i *row, *maxRow, *maxCol, *lastCol;
    posix_memalign((void **)&sp, 32, (size_t)n * 23 * 16 + 64);
    posix_memalign((void **)&row, 32, (size_t)(block + 1) * sizeof(__m128i));
    posix_memalign((void **)&maxCol, 32, (size_t)(block + 1) * sizeof(__m128i));
    posix_memalign((void **)&maxRow, 32, (size_t)(m + 1) * sizeof(__m128i));
    posix_memalign((void **)&lastCol, 32, (size_t)(m + 1) * sizeof(__m128i));
    char *bpad;
    posix_memalign((void **)&bpad, 32, (size_t)n * 16 + 64);
    memcpy(bpad, b16, (size_t)n * 16);
    memset(bpad + (size_t)n * 16, 23, 64);
    sw_host((char *)a, m, bpad, n, (char *)submat, scores16, overflow, (char)open_gap, (char)extend_gap, sp, row, maxRow, maxCol, lastCol);
    free(sp); free(row); free(maxCol); free(maxRow); free(lastCol); free(bpad);
}

/* The reference's descending sort; titles are used to carry the original
 * indices so that the permutation (the tie order) can be read back. */
void ref_sort_scores(int *scores, uint32_t *index_out, unsigned long size, int threads)
{
    char **titles = (char **)malloc((size ? size : 1) * sizeof(char *));
    for (unsigned long i = 0; i < size; i++) titles[i] = (char *)(uintptr_t)(i + 1);
    sort_scores(scores, titles, size, threads);
    for (unsigned long i = 0; i < size; i++) index_out[i] = (uint32_t)((uintptr_t)titles[i] - 1);
    free(titles);
}

void ref_preprocess_db(const char *input_filename, const char *out_filename, int threads)
{
    preprocess_db((char *)input_filename, (char *)out_filename, threads);
}

/* Query loader: returns counts; the arrays are handed out through pointers
 * that stay owned by this shim until ref_free_queries(). */
static char *g_a; static char **g_titles; static unsigned short *g_m; static unsigned int *g_disp;
static unsigned long g_nq, g_Q;

unsigned long ref_load_queries(const char *filename, int threads, unsigned long *Q)
{
    load_query_sequences((char *)filename, &g_a, &g_titles, &g_m, &g_nq, &g_Q, &g_disp, threads);
    *Q = g_Q;
    return g_nq;
}
const char *ref_queries_residues(void) { return g_a; }
const unsigned short *ref_queries_lengths(void) { return g_m; }
const unsigned int *ref_queries_disp(void) { return g_disp; }
const char *ref_queries_title(unsigned long i) { return g_titles[i]; }

/* Chunk assembly for the device layout (a4). */
static char **c_b; static unsigned int c_count; static unsigned int *c_groups; static unsigned long *c_accum, *c_vD;
static unsigned short **c_n, **c_nbb; static unsigned int **c_disp;
static unsigned long c_seqs, c_D, c_vgroups, c_vDtot, c_maxvD; static unsigned short c_maxlen; static int c_maxtitle;

unsigned int ref_assemble(const char *dbname, int vector_length, unsigned long max_chunk, unsigned int ndev, int threads,
                          unsigned long *out8 /* seqs, D, maxlen, maxtitle, vgroups, vD, max_chunk_vD, chunk_count */)
{
    assemble_multiple_chunks_db((char *)dbname, vector_length, max_chunk, ndev, &c_seqs, &c_D, &c_maxlen, &c_maxtitle, &c_vgroups, &c_vDtot,
                                &c_b, &c_count, &c_groups, &c_accum, &c_vD, &c_maxvD, &c_n, &c_nbb, &c_disp, threads);
    out8[0] = c_seqs; out8[1] = c_D; out8[2] = c_maxlen; out8[3] = (unsigned long)c_maxtitle;
    out8[4] = c_vgroups; out8[5] = c_vDtot; out8[6] = c_maxvD; out8[7] = c_count;
    return c_count;
}
unsigned int ref_chunk_groups(unsigned int c) { return c_groups[c]; }
unsigned long ref_chunk_accum(unsigned int c) { return c_accum[c]; }
unsigned long ref_chunk_vD(unsigned int c) { return c_vD[c]; }
const char *ref_chunk_b(unsigned int c) { return c_b[c]; }
const unsigned short *ref_chunk_n(unsigned int c) { return c_n[c]; }
const unsigned short *ref_chunk_nbb(unsigned int c) { return c_nbb[c]; }
const unsigned int *ref_chunk_disp(unsigned int c) { return c_disp[c]; }

/* Headers as the reference reads them back for the report. */
static char **h_titles; static unsigned long h_count;
void ref_load_headers(const char *dbname, unsigned long count, int max_title_length)
{
    load_database_headers((char *)dbname, count, max_title_length, &h_titles);
    h_count = count;
}
const char *ref_header(unsigned long i) { return h_titles[i]; }

} /* extern "C" */
