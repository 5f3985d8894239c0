/*
 * phyloligo_amd.h -- C ABI of the MI355X-native PhylOligo all-by-all contig distance path.
 *
 * The reference (itsmeludo/PhylOligo) is pure Python and has no FFI of its own; its seam for
 * this path is two dispatcher functions keyed by the `--method` string plus a metric-name
 * registry (paths relative to /root/reference/phylopackage/):
 *
 *   compute_frequencies(mthdrun, large, genome, pattern, strand, ...)   bin/phyloligo.py:980-997
 *   compute_distances(mthdrun, large, frequencies, ..., dist, ...)      bin/phyloligo.py:536-553
 *   call_dist = {"Eucl","JSD","KT","BC","SC"}                           bin/phyloligo.py:381
 *   numpy.savetxt(out_file, res, delimiter="\t")                        bin/phyloligo.py:1059-1066
 *
 * Each entry point below names the reference function(s) it replaces.  The binding a
 * PhylOligo maintainer would add (a ctypes stub selected by `--method hip`) is shown in
 * INTEGRATION.md.  Plain pointers and sizes only; no torch / numpy types cross this line.
 *
 * Conventions
 *   - every function returns PO_OK (0) or a negative po_status; po_last_error() gives the
 *     message of the calling thread's last failure;
 *   - `*_dev` entry points take DEVICE pointers and enqueue on the context's stream without
 *     synchronising (the caller owns the stream: po_ctx_set_stream / po_ctx_synchronize);
 *     the un-suffixed forms take HOST pointers, copy in, run the same kernels, copy out and
 *     return when the result is in the caller's buffer;
 *   - a profile is an exact integer count vector in the reference's word order
 *     (itertools.product("CGAT", repeat=k), bin/phyloligo.py:653: C=0,G=1,A=2,T=3, first
 *     letter most significant) plus the number of counted words of the record;
 *   - there is NO CPU fallback: without a HIP device every compute call fails with PO_ENODEV.
 */
#ifndef PHYLOLIGO_AMD_H
#define PHYLOLIGO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_ABI_VERSION 1

typedef struct po_ctx po_ctx;

typedef enum po_status {
    PO_OK = 0,
    PO_EINVAL = -1,       /* bad argument (unknown strand / metric / pattern, null pointer, bad range) */
    PO_ENODEV = -2,       /* no usable HIP device */
    PO_ENOMEM = -3,       /* host or device allocation failed */
    PO_EHIP = -4,         /* a HIP runtime call or kernel launch failed */
    PO_EUNSUPPORTED = -5, /* valid request outside the implemented envelope (window > 64, k > 8) */
    PO_EIO = -6,          /* file could not be opened / written, malformed FASTA */
} po_status;

/* -s/--strand choices, bin/phyloligo.py:1010 and select_strand :124-149 */
typedef enum po_strand { PO_STRAND_BOTH = 0, PO_STRAND_PLUS = 1, PO_STRAND_MINUS = 2 } po_strand;

/* -d/--distance choices, bin/phyloligo.py:1012; functions in core/phylodist.py:36-85 */
typedef enum po_metric { PO_EUCL = 0, PO_JSD = 1, PO_KT = 2, PO_BC = 3, PO_SC = 4,
                         PO_KL = 5 /* Kount.py only (po_profile_distances), bin/Kount.py:69-85 */ } po_metric;

/* element type of the distance matrix: float64 is what compute_distances_joblib returns
 * (bin/phyloligo.py:364-392); float32 is the container type of the --large memmap variant
 * (bin/phyloligo.py:413) -- values are computed in float64 and rounded once on store. */
typedef enum po_dtype { PO_F64 = 0, PO_F32 = 1 } po_dtype;

/* flags of po_pairwise* */
#define PO_FLAG_NO_SYMMETRY 1u /* compute every (i,j) independently even for the full matrix
                                  (what sklearn does for n_jobs>1); default mirrors j<i from i<j */

#define PO_FLAG_NO_TABLE_PATH 2u /* general kernels only: no integer-sum table kernel for JSD record blocks with
                                   equal word totals, no exact int8-MFMA kernel for Eucl / SC,
                                   no packed-byte SAD kernel for BC, no int8-MFMA kernel for KT, and frequency
                                   input is not traced back to integer profiles (po_pairwise_freq*: by default
                                   a matrix whose every entry is count / total bit for bit, i.e. count2freq output,
                                   is; that check reads one flag word back)                              */

#define PO_FLAG_NO_RC_FOLD 4u /* JSD / BC / KT: do not look for reverse-complement symmetric profiles.  By default
                                 the input is checked on the device (count[w] == count[rc(w)] for every record
                                 and word - what `-s both` with a palindromic pattern produces,
                                 bin/phyloligo.py:141) and, if it holds, the sums over words run over one word
                                 per {w, rc(w)} orbit: same result up to summation order, about half
                                 the work.  The check reads one flag word back, i.e. it synchronises the
                                 stream once per call; this flag avoids that.                             */

#define PO_FLAG_PAIRDOT_I8 8u /* KT (dim <= 256) and BC on thermometer planes: keep the materialised {-1,0,1} operand as
                                int8 and use v_mfma_i32_32x32x32_i8 instead of the default FP4 (E2M1) operand with
                                v_mfma_scale_f32_32x32x64_f8f6f4.  Both are exact; results are bit-identical.   */

#define PO_FLAG_NO_PAIRDOT 16u /* KT / BC: do not materialise pair-sign / thermometer operands for the matrix cores; BC then
                                 takes the packed-byte SAD kernel (equal-total blocks) and the general kernel, KT at
                                 dim <= 256 the O(D^2) vector kernel.  For cross-checks and timing comparisons.      */

/* Filled by po_pairwise* when non-NULL.  Times are HIP-event times on the context's stream;
 * asking for them makes the call synchronise. */
typedef struct po_stats {
    double prep_ms;        /* counts -> device working layout (+ per-row terms)                  */
    double kernel_ms;      /* the tile kernel(s) of the metric                                    */
    double total_ms;       /* prep + kernel + anything between                                    */
    uint64_t pairs;        /* matrix entries written / 2 (unordered pairs incl. half the diagonal) */
    uint64_t tiles;        /* workgroup tiles launched                                            */
    uint32_t kernel_id;    /* which tile kernel ran (PO_KERNEL_*)                                 */
    uint32_t rc_folded;    /* 1 if the reverse-complement folded operands were used (PO_FLAG_NO_RC_FOLD)  */
} po_stats;

#define PO_KERNEL_VALU_JSD 1u
#define PO_KERNEL_VALU_BC 2u
#define PO_KERNEL_MFMA_F64_GRAM 3u
#define PO_KERNEL_MFMA_I8_GRAM 4u  /* exact int8 kernels (counts <= 2097151, dim <= 32768; Spearman ranks, dim <= 16384), else the float64 one */
#define PO_KERNEL_VALU_KT 5u
#define PO_KERNEL_MFMA_I8_KT 8u /* Kendall tau as an exact matrix-core Gram over pair-sign vectors (FP4 / int8 operands) */
#define PO_KERNEL_MFMA_BC 9u    /* Bray-Curtis as s_a + s_b - 2 <thermometer(a), thermometer(b)> on the matrix cores   */
#define PO_KERNEL_SAD_BC 7u    /* packed-byte SAD kernel (equal-total blocks) + general kernel for the rest */
#define PO_KERNEL_LUT_JSD 6u   /* integer-sum table kernel (equal-total blocks, counts <= 255) + general kernel for the remaining tiles */

/* ---- library / context ------------------------------------------------------------------ */
const char* po_version(void);   /* "phyloligo_amd 0.1 (gfx950) src <16 hex digits>": the hash of the sources it was built from */
int po_abi_version(void);
const char* po_last_error(void);
const char* po_status_string(int status);
int po_device_count(void);                              /* 0 when no HIP device is visible      */
int po_ctx_create(po_ctx** out, int device_id);         /* one context per GPU / per process rank */
void po_ctx_destroy(po_ctx* ctx);
int po_ctx_set_stream(po_ctx* ctx, void* hip_stream);   /* hipStream_t of the caller; NULL = default */
int po_ctx_synchronize(po_ctx* ctx);
int po_ctx_device_name(po_ctx* ctx, char* buf, size_t len);
/* Frees every device workspace the context has grown (operand matrices, the materialised Kendall / Bray-Curtis
 * operand of up to 96 GB, staging of the host-pointer forms); the next call allocates again.  The reference's
 * workers hold nothing between calls (joblib processes, bin/phyloligo.py:386-390): this is the way back to that. */
int po_ctx_trim(po_ctx* ctx);

/* ---- pattern ---------------------------------------------------------------------------- *
 * `pattern` is the -p string of '1'/'0' (bin/phyloligo.py:1027); -k N is "1"*N (:1040-1041).
 * window = len(pattern) <= 64, k = number of '1' <= 8, dim = 4^k.  (The reference takes any length and any k,
 * bin/phyloligo.py:622-628; 4^9 columns and beyond are impractical anywhere, wider seeds than 64 return
 * PO_EUNSUPPORTED.)                                                                           */
int po_pattern_info(const char* pattern, uint32_t* window, uint32_t* k, uint64_t* dim);

/* ---- stage 1: profiles ------------------------------------------------------------------ *
 * Replaces compute_frequencies_joblib (bin/phyloligo.py:847-877), i.e. per record
 * select_strand (:124-149) -> upper() (:683) -> cut_sequence_and_count_pattern (:601-631) ->
 * the dense C,G,A,T ordering of count2freq (:653); the division count/total of :656 is left
 * to the consumer (po_frequencies*, po_pairwise*) so that the canonical result is exact.
 *   seq      concatenated sequence bytes of all records, in file order, line ends / blanks
 *            already removed, case preserved (any byte that is not ACGTacgt separates words)
 *   offsets  n_seqs+1 byte offsets into seq (offsets[0]=0, non-decreasing)
 *   counts   [n_seqs][dim] uint32, row major          totals  [n_seqs] uint64                 */
int po_count_profiles(po_ctx* ctx, const uint8_t* seq, const uint64_t* offsets, uint64_t n_seqs,
                      const char* pattern, int strand, uint32_t* counts, uint64_t* totals);
int po_count_profiles_dev(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t n_seqs,
                          uint64_t total_bytes, const char* pattern, int strand,
                          uint32_t* d_counts, uint64_t* d_totals);

/* count2freq (bin/phyloligo.py:633-661): freq[i][w] = counts[i][w] / totals[i] in float64
 * (0 for an empty record); this is the matrix -q/--outfreq writes (:1059-1061).               */
int po_frequencies(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                   double* freq);
int po_frequencies_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                       uint32_t dim, double* d_freq);

/* ---- sliding windows against a prototype (bin/Kount.py, the ContaLocate front end) ----------- *
 * Profiles of arbitrary, possibly overlapping byte ranges [begins[i], ends[i]) of one sequence buffer:
 * the windows that make_genome_chunk cuts (bin/Kount.py:343-407), each counted exactly like a record
 * (cut_sequence_and_count_pattern with the strand handling inside, bin/Kount.py:208-243).
 * sum_lengths (device form) is an upper bound of sum(ends[i]-begins[i]).                          */
int po_count_profiles_ranges(po_ctx* ctx, const uint8_t* seq, uint64_t total_bytes, const uint64_t* begins,
                             const uint64_t* ends, uint64_t n_ranges, const char* pattern, int strand,
                             uint32_t* counts, uint64_t* totals);
int po_count_profiles_ranges_dev(po_ctx* ctx, const uint8_t* d_seq, uint64_t total_bytes, const uint64_t* d_begins,
                                 const uint64_t* d_ends, uint64_t n_ranges, uint64_t sum_lengths, const char* pattern,
                                 int strand, uint32_t* d_counts, uint64_t* d_totals);
/* Distance of every profile to ONE prototype frequency vector proto[dim] (compute_distance_joblib,
 * bin/Kount.py:322-330): PO_JSD / PO_EUCL / PO_KL of bin/Kount.py:69-123 WITHOUT their x1000 display
 * scaling (the host mirror applies it).  out[n] float64.                                           */
int po_profile_distances(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                         const double* proto, int metric, double* out);
int po_profile_distances_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                             uint32_t dim, const double* d_proto, int metric, double* d_out);
/* Occurrences of one byte value in every range - the numerator of Kount.py's N gate, seq.count("N") / len(seq)
 * (bin/Kount.py:295), for the same windows, without a pass over the genome on the host.  Device pointers.       */
int po_count_byte_ranges_dev(po_ctx* ctx, const uint8_t* d_seq, uint64_t total_bytes, const uint64_t* d_begins,
                             const uint64_t* d_ends, uint64_t n_ranges, int byte, uint64_t* d_out);

/* ---- stage 2: pairwise matrix ----------------------------------------------------------- *
 * Replaces compute_distances_joblib (bin/phyloligo.py:364-392) = sklearn pairwise_distances
 * over phylodist.Eucl / JSD / KT / SC (core/phylodist.py:36-85) and SciPy 'braycurtis'.
 * Computes rows [row_begin,row_end) x all n columns:
 *     out[(i-row_begin)*ld_out + j]   0 <= j < n,   ld_out >= n  (elements, not bytes)
 * Diagonal as the reference produces it: Eucl/JSD/BC/SC 0, KT 1 (0 for a constant row).
 * Row blocks are independent, which is how the matrix shards over GPUs (one context each).
 * Layout of a DEVICE result and speed: any ld_out >= n is correct.  Rows that start on 16-byte boundaries (d_out 16-byte aligned and
 * ld_out a multiple of 4 float32 / 2 float64 entries) leave as 16-byte stores; rows on whole 128-byte lines (ld_out a multiple of 32
 * float32 / 16 float64 entries) are written ~25 % faster still, and an odd float32 leading dimension costs about 2 x (every 512-byte
 * row piece of a tile then begins and ends inside a 32-byte sector that a neighbouring tile also writes).  The host-pointer forms
 * below keep their device copy of the result on 128-byte rows whatever n is.  The metric of the largest matrices here, Eucl on the
 * exact int8 path, makes ONE host synchronisation per call from 8 192 records on (it reads 4 bytes per 128 records back to deal the
 * tiles to the kernels of their class); JSD and BC make one for the fold decision, as before.                                        */
int po_pairwise(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                int metric, uint64_t row_begin, uint64_t row_end, int out_dtype, void* out, uint64_t ld_out,
                uint32_t flags, po_stats* stats);
int po_pairwise_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                    int metric, uint64_t row_begin, uint64_t row_end, int out_dtype, void* d_out,
                    uint64_t ld_out, uint32_t flags, po_stats* stats);

/* The same from a float64 frequency matrix freq[n][dim] (row major) -- literally the
 * `frequencies` argument of compute_distances / compute_distances_joblib
 * (bin/phyloligo.py:536-553, :364-392), for callers that hold frequencies rather than counts. */
int po_pairwise_freq(po_ctx* ctx, const double* freq, uint64_t n, uint32_t dim, int metric, uint64_t row_begin,
                     uint64_t row_end, int out_dtype, void* out, uint64_t ld_out, uint32_t flags, po_stats* stats);
int po_pairwise_freq_dev(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, int metric,
                         uint64_t row_begin, uint64_t row_end, int out_dtype, void* d_out, uint64_t ld_out,
                         uint32_t flags, po_stats* stats);

/* Several rectangular blocks of one matrix in one call (the per-rank work list of a multi-GPU run):
 * the working layout is prepared once, then every block is launched.
 *   out[(i-row_begin)*ld_out + (j-col_begin)]                      row_begin <= i < row_end, col_begin <= j < col_end
 *   mirror[(j-col_begin)*ld_mirror + (i-row_begin)]  (if not NULL)  the transposed block, same values
 * triangular != 0 requires rows == columns: only pairs i <= j are evaluated and the lower half is
 * filled by symmetry inside `out` (mirror / ld_mirror are ignored).                                    */
typedef struct po_block {
    uint64_t row_begin, row_end, col_begin, col_end;
    void* out;
    uint64_t ld_out;
    void* mirror;
    uint64_t ld_mirror;
    uint32_t triangular;
    uint32_t reserved;
} po_block;
int po_pairwise_blocks_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                           int metric, int out_dtype, const po_block* blocks, uint32_t n_blocks, uint32_t flags,
                           po_stats* stats);

/* bytes of device workspace po_pairwise_dev will hold for this problem (allocated lazily on
 * first use and kept by the context; call once before timing to keep hipMalloc out of it)    */
int po_pairwise_reserve(po_ctx* ctx, uint64_t n, uint32_t dim, int metric);

/* ---- host-side formats either side of the path ------------------------------------------ *
 * FASTA ingest with the semantics of Bio.SeqIO.parse(genome, "fasta") as used at
 * bin/phyloligo.py:869: '>' at line start opens a record, sequence lines are right-stripped
 * and joined, ' ' and '\r' removed.  Two calls: sizes first, then fill.                      */
int po_fasta_scan(const uint8_t* data, uint64_t len, uint64_t* n_records, uint64_t* seq_bytes);
int po_fasta_extract(const uint8_t* data, uint64_t len, uint8_t* seq_out, uint64_t* offsets_out,
                     uint64_t* title_begin, uint64_t* title_end);

/* ---- FASTA ingest on the device ------------------------------------------------------------ *
 * The same records as po_fasta_scan / po_fasta_extract, from the raw file bytes already in HBM (16-byte aligned
 * buffer): what Bio.SeqIO.parse(genome, "fasta") yields at bin/phyloligo.py:869, without a pass over the file on
 * the host.  po_fasta_scan_dev sizes the outputs (it synchronises the stream once) and must precede
 * po_fasta_extract_dev on the same buffer.  Title spans [title_begin, title_end) end at the line end: strip trailing
 * white space when decoding a title.  Returns PO_EIO for text before the first record, PO_EUNSUPPORTED for a tab /
 * vertical tab / form feed on a sequence line (rstrip() semantics that need the host parser).                   */
int po_file_read(const char* path, uint8_t* buf, uint64_t len);   /* first len bytes of a file, read in parallel (host) */
int po_fasta_scan_dev(po_ctx* ctx, const uint8_t* d_data, uint64_t len, uint64_t* n_records, uint64_t* seq_bytes);
int po_fasta_extract_dev(po_ctx* ctx, const uint8_t* d_data, uint64_t len, uint8_t* d_seq, uint64_t* d_offsets,
                         uint64_t* d_title_begin, uint64_t* d_title_end);

/* numpy.savetxt(path, m, delimiter="\t") of bin/phyloligo.py:1061,1066: "%.18e" values, '\t'
 * between columns, '\n' after each row, "nan"/"inf" spelled as numpy spells them.  A regular file that exists is
 * overwritten in place and cut to the new length at the end (append = 1: continued at its end); the rows are formatted and
 * written by the host threads the job may use.  Any other kind of file (pipe, /dev/null) gets the bytes in order.          */
int po_write_mat_text(const double* m, uint64_t rows, uint64_t cols, uint64_t ld, const char* path, int append);

/* The write side of the raw float32 container of `--large memmap` (bin/phyloligo.py:394-427: row slices assigned into a
 * numpy.memmap of the output file, :200-217; read back by phyloligo_comparemat.py:16-24 and phyloselect.py:606-614):
 * `rows` pieces of row_bytes bytes, piece r taken from src + r * src_pitch, go to byte offset file_offset + r * file_pitch
 * of the OPEN file descriptor fd through up to `threads` parallel pwrite(2) callers (<= 0: 8; never more than the CPUs the
 * job may use).  Whole rows of the matrix (row_bytes == src_pitch == file_pitch) are written as one contiguous range; a
 * rectangular block of a multi-GPU work list (some columns of some rows) is one pwrite per row.  Host only: no po_ctx, no
 * GPU.  Several processes may write disjoint ranges of one file at once.  Returns PO_EIO with the errno text on failure. */
int po_pwrite_rows(int fd, const void* src, uint64_t rows, uint64_t row_bytes, uint64_t src_pitch,
                   uint64_t file_offset, uint64_t file_pitch, int threads);

#ifdef __cplusplus
}
#endif
#endif /* PHYLOLIGO_AMD_H */
