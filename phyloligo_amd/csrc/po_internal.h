// Internal declarations shared by the translation units of libphyloligo_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "phyloligo_amd.h"
#include "po_host.h"

// ---- error plumbing --------------------------------------------------------------------
void po_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define PO_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            po_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (e__ == hipErrorOutOfMemory) ? PO_ENOMEM : PO_EHIP;                     \
        }                                                                                  \
    } while (0)

#define PO_CHECK_LAUNCH(name)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            po_set_error("launch of %s failed: %s", name, hipGetErrorString(e__));         \
            return PO_EHIP;                                                                \
        }                                                                                  \
    } while (0)

#define PO_REQUIRE(cond, ...)                                                              \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            po_set_error(__VA_ARGS__);                                                     \
            return PO_EINVAL;                                                              \
        }                                                                                  \
    } while (0)

// ---- context ---------------------------------------------------------------------------
// A growable device buffer owned by the context: it only grows between po_ctx_trim / po_ctx_destroy (and the pair-dot operand
// is released when a call for another metric follows), so the steady state of repeated calls does no hipMalloc.
struct po_buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct po_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // caller's stream (not owned) or the default stream
    hipDeviceProp_t prop;
    po_buf ws_freq;                    // Ft: float64 [dim][npad] transposed frequencies
    po_buf ws_rowstat;                 // per-row terms (entropy, norm, ...), float64 [4][npad]
    po_buf ws_aux;                     // metric specific (ranks, sign expansions, chunk tables)
    po_buf ws_io;                      // staging of the host-pointer entry points
    po_buf ws_logtab;                  // replicated log table of the JSD kernel
    bool logtab_ready = false;
    po_buf ws_fold;                    // reverse-complement folded counts / frequencies (po_fold.hip) + flag word
    po_buf ws_fold_src;                // source word of every folded column, for (fold_dim, fold_gran)
    po_buf ws_recover;                 // integer profiles recovered from a frequency matrix (po_recover.hip)
    po_buf ws_pairdot;                 // materialised operand of the pair-dot kernels (po_pairdot.hip)
    po_buf ws_pq;                      // Kendall's word-pair table, cached per (dim, fold layout, format)
    uint64_t pq_key = ~0ull;
    po_buf ws_thermo;                  // Bray-Curtis thermometer plan (levels per word, element map)
    po_buf ws_seg;                     // stage 1: segment rows of long records (po_count.hip), all zero between calls
    po_buf ws_scan;                    // stage 1: per-workgroup aggregates of the one-launch chunk scan, tagged with scan_epoch
    uint32_t scan_epoch = 0;           // (entries of older calls carry older tags: the buffer is never cleared between calls)
    po_buf ws_fasta;                   // per-block partial results of the on-device FASTA scan (po_fasta.hip)
    const uint8_t* fasta_data = nullptr;   // the buffer po_fasta_scan_dev last sized, with its length and totals
    uint64_t fasta_len = 0, fasta_records = 0, fasta_seq_bytes = 0;
    uint32_t fold_dim = 0, fold_gran = 0, fold_dim_f = 0, fold_dbl_at = 0;
    uint32_t* h_flag = nullptr;        // pinned host word for the fold decision
    uint32_t* h_blockmax = nullptr;    // pinned host copy of the int8 Gram path's largest counts (whole matrix, then per 128-record block)
    size_t h_blockmax_cap = 0;         // in words
    po_buf ws_tilelist;                // per-class tile lists of the int8 Gram path (po_gram_i8.hip)
    po_buf ws_fix;                     // po_fix_list of the JSD path (po_jsd_exact.hip)
    void* h_stage[2] = {nullptr, nullptr};   // pinned staging buffers of the host-pointer entry points (device -> host rows)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

int po_buf_reserve(po_ctx* ctx, po_buf* b, size_t bytes);
// hipFuncAttributeMaxDynamicSharedMemorySize for `func`, raised once per process and device (and again only if a
// later launch asks for more; it never goes down): the attribute call is host work that does not belong in a launch path
int po_func_shmem(po_ctx* ctx, const void* func, size_t bytes);
#define PO_SHMEM(ctx, kernel, bytes)                                                       \
    do {                                                                                   \
        int rc__ = po_func_shmem((ctx), reinterpret_cast<const void*>(kernel), (bytes));   \
        if (rc__) return rc__;                                                             \
    } while (0)

// ---- pattern ---------------------------------------------------------------------------
#define PO_MAX_WINDOW 64
#define PO_MAX_K 8
#define PO_MAX_RUNS 16

// A spaced-word pattern compiled for the rolling 2-bit window register: the word index is
// the OR over runs of ((reg >> src_shift) & mask) << dst_shift.
struct po_pattern {
    uint32_t window;   // W = len(pattern)
    uint32_t k;        // number of '1'
    uint32_t dim;      // 4^k
    uint32_t nruns;
    uint32_t src_shift[PO_MAX_RUNS];
    uint32_t dst_shift[PO_MAX_RUNS];
    uint32_t mask[PO_MAX_RUNS];
    uint32_t ones[PO_MAX_WINDOW];  // positions of the '1's (junction words are built from these)
};
int po_pattern_compile(const char* pattern, po_pattern* out);

// ---- kernels (one launcher per translation unit) -----------------------------------------
// record i = bytes [d_begins[i], d_ends[i]) of d_seq; sum_lengths >= sum of the record lengths
int po_launch_count(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_begins, const uint64_t* d_ends, uint64_t n_seqs,
                    uint64_t total_bytes, uint64_t sum_lengths, const po_pattern& pat, int strand, uint32_t* d_counts,
                    uint64_t* d_totals);
// distance of every profile to one prototype frequency vector (Kount.py); metric PO_EUCL / PO_JSD / PO_KL
int po_launch_profile_distances(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                                const double* d_proto, int metric, double* d_out);

int po_launch_count_byte_ranges(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_begins, const uint64_t* d_ends, uint64_t n,
                                uint32_t byte, uint64_t* d_out);

// Working layout of stage 2: Ft[d][npad] = counts[n][d] / totals[n] (float64, zero padded).
// skip_flag (may be NULL): device word with the largest count; the kernel does nothing when it is <= skip_upto
int po_launch_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                   uint64_t npad, double* d_ft, const uint32_t* skip_flag, uint32_t skip_upto);
int po_launch_prep_freq(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, uint64_t npad, double* d_ft);
int po_launch_freq_rowmajor(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                            uint32_t dim, double* d_freq);
// rowstat[0][n] = sum f ln f, rowstat[1][n] = sum f
// logtab (may be NULL): the JSD log table; when given, rowstat[2] = sum f ln f by the tile kernel's table log
int po_launch_rowstat(po_ctx* ctx, const double* d_ft, uint64_t n, uint32_t dim, uint64_t npad, double* d_rowstat,
                      const void* logtab, uint32_t dbl_at);

// frequency matrix -> the integer profiles it was made from, verified bit for bit (po_recover.hip)
int po_recover_counts(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, bool* recovered,
                      const uint32_t** d_counts, const uint64_t** d_totals);

// rowstat[0] (sum f ln f, when asked) and rowstat[1] (sum f) straight from integer counts: used instead of the float64
// operand matrix + po_launch_rowstat when the equal-total kernels are known to own every tile
int po_launch_rowstat_counts(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                             uint64_t npad, double* d_rowstat, bool want_entropy, uint32_t dbl_at);

// One rectangular block of the matrix handed to a tile kernel.
struct po_tile_args {
    const double* ft;       // operand matrix [dim8][npad] (frequencies or centred ranks)
    const double* rowstat;  // [4][npad]
    uint64_t n, npad;
    uint32_t dim;
    uint64_t row_begin, row_end;   // rows of the block (absolute record indices)
    uint64_t col_begin, col_end;   // columns of the block
    void* out;              // out[(i-row_begin)*ld_out + (j-col_begin)]
    uint64_t ld_out;
    void* mirror;           // optional transpose target: mirror[(j-col_begin)*ld_mirror + (i-row_begin)]
    uint64_t ld_mirror;
    int out_f32;
    int triangular;         // rows == columns: only tiles on/above the diagonal are computed (mirror = out)
    uint32_t dbl_at;        // word index at which running sums double (reverse-complement folded operands,
                            // po_fold.hip); 0xFFFFFFFF = never.  A multiple of the kernel's staging step.
    struct po_fix_list* fix;  // JSD: where a wave notes that its rows of a tile hold a value at cancellation level (or NULL)
};
// JSD of near-identical records (po_jsd_exact.hip).  1/2 (E_a + E_b - S) subtracts sums of size ln D and is good to ~1e-14 absolute;
// a wave of a JSD tile kernel whose rows hold a value below 2^-20 appends (tile, first row, rows) here, and a small kernel behind the
// tile kernels evaluates exactly those values again, word by word in a cancellation-free form.  More entries than the list holds:
// none is processed (all or nothing, so that a result never depends on the order of the appends).
#define PO_FIX_CAP 16384u
#define PO_FIX_BELOW_HI 0x3EB00000u    // high word of 2^-20: v >= 0 is below 2^-20 iff hi(v) < this
struct po_fix_list {
    uint32_t count, pad;
    unsigned long long entry[PO_FIX_CAP];     // ti << 40 | tj << 16 | first row << 8 | rows   (ti, tj < 2^24)
};
#define PO_NO_DOUBLING 0xFFFFFFFFu
#define PO_FOLD_SELFS_FIRST 0xFFFFu    // `gran` value of po_rc_fold selecting the Kendall layout [self-paired | representatives]
// Reverse-complement folding of strand-symmetric inputs (JSD, BC); see po_fold.hip
uint32_t po_fold_selfs(uint32_t dim);   // self-paired words of a 4^k word space (0xFFFFFFFF: dim is not 4^k)
#define PO_FOLD_ASYM 1u            // flag bits of the fold pass: some record is not reverse-complement symmetric
#define PO_FOLD_NOT_ALL_TABLE 2u   // some 128-record block does not qualify for the JSD integer-sum table kernel
#define PO_FOLD_NOT_ALL_SAD 4u     // some 128-record block does not qualify for the packed SAD kernel (BC)
#define PO_FOLD_SOME_EQUAL 8u      // some record shares its (non-zero) total with seven other records of its 128-record block: a block
                                   // with one common total MAY exist.  Clear = certainly none does: a ragged assembly, where the
                                   // equal-total kernels (JSD table, BC SAD / thermometer) would be launched to own no tile at all
int po_rc_fold(po_ctx* ctx, const uint32_t* d_counts, const double* d_freq, const uint64_t* d_totals, uint64_t n,
               uint32_t dim, uint32_t gran, bool* folded, uint32_t* dim_f, uint32_t* dbl_at, uint32_t* flags_out);
// cls (may be NULL): per 128-record block, the common word total if the equal-total table path owns
// the tiles of that class (po_jsd_lut.hip); valu_tile_kernel<JSD> skips tiles with equal non-zero classes.
int po_launch_valu_tiles(po_ctx* ctx, int metric, const po_tile_args& a, const unsigned long long* cls, uint64_t* tiles);
size_t po_jsd_lut_workspace(uint64_t n, uint32_t dim);
// d_wsum: rowstat[1] = sum_w f of every record (po_launch_rowstat must have run)
int po_launch_jsd_lut_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                           uint64_t npad, const double* d_wsum, void* ws, const unsigned long long** cls_out);
int po_launch_jsd_lut_tiles(po_ctx* ctx, const po_tile_args& a, uint64_t n, const void* ws, uint64_t* tiles);
// po_jsd_exact.hip: the list the JSD tile kernels append to (emptied), and the pass behind them
int po_jsd_exact_reset(po_ctx* ctx, po_fix_list** list);
int po_launch_jsd_exact(po_ctx* ctx, const po_tile_args& a, const uint32_t* d_counts, const uint64_t* d_totals, const double* d_freq, uint32_t dim);
size_t po_bc_sad_workspace(uint64_t n, uint32_t dim);
int po_launch_bc_sad_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                          uint64_t npad, void* ws, const unsigned long long** cls_out);
int po_launch_bc_sad_tiles(po_ctx* ctx, const po_tile_args& a, const void* ws, uint64_t* tiles);
size_t po_gram_i8_workspace(uint64_t n, uint32_t dim);
uint32_t po_gram_i8_value_limit(uint32_t dim);       // largest count the exact int8 kernels take (three 7-bit digits where the accumulators hold them)
bool po_gram_i8_sc_supported(uint32_t dim);          // Spearman's doubled centred ranks fit two digits
// digit planes + per-record terms from uint32 counts (Eucl) or int32 doubled centred ranks (SC, signed_values)
int po_launch_gram_i8_prep(po_ctx* ctx, const uint32_t* d_vals, const uint64_t* d_totals, bool signed_values, uint64_t n,
                           uint32_t dim, uint64_t npad, void* ws, const uint32_t** maxabs_out);
// h_max (may be NULL): host copy of the largest counts, from po_gram_i8_block_maxima - the launcher then deals the tiles to the
// one- / two- / three-plane kernels by the classes of their two record blocks (exact grids, nothing launched to exit at once);
// NULL: every candidate kernel is launched over all tiles and the device-side maximum decides which one runs
int po_launch_gram_i8_tiles(po_ctx* ctx, int metric, const po_tile_args& a, const void* ws, const uint32_t* h_max, uint64_t* tiles);
// waits for the stream: *h_max = [largest count of the matrix, largest count of every block of 128 records]
int po_gram_i8_block_maxima(po_ctx* ctx, const void* ws, uint64_t npad, uint32_t dim, const uint32_t** h_max);
int po_launch_gram_norms(po_ctx* ctx, const double* ft, uint32_t dim, uint64_t npad, double* rowstat, const uint32_t* skip_flag,
                         uint32_t skip_upto);
// i8flag (may be NULL): device word holding the largest count; the float64 kernel leaves the matrix to the
// int8 kernels when it is <= i8_upto
int po_launch_gram_f64(po_ctx* ctx, int metric, const po_tile_args& a, const uint32_t* i8flag, uint32_t i8_upto, uint64_t* tiles);
int po_logtab_init(po_ctx* ctx);

// KT / SC helpers.  Order statistics of every record, from counts (uint32) or frequencies (float64):
//   d_rt        centred average ranks, float64 [dim][npad]   (may be NULL)
//   d_lessrank  number of strictly smaller words, uint32 [n][dim]: same order and ties as the
//               input, which is all Kendall's tau looks at            (may be NULL)
//   d_r2        2 #less + #equal - dim (twice the centred average rank), int32 [n][dim]   (may be NULL)
//   rowstat[3]  number of tied word pairs of the record
int po_launch_ranks(po_ctx* ctx, const uint32_t* d_counts, const double* d_freq, uint64_t n, uint32_t dim,
                    uint64_t npad, double* d_rt, uint32_t* d_lessrank, int32_t* d_r2, double* d_rowstat);
int po_launch_kt(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, const po_tile_args& a,
                 uint64_t* tiles);

// Kendall / Bray-Curtis as exact Gram matrices over a materialised {-1,0,1} operand (po_pairdot.hip)
struct po_pairdot_plan {
    int fmt_fp4;            // operand format: 0 = int8 (16 elements per 16-byte chunk), 1 = FP4 E2M1 (32)
    uint32_t n_stages;      // stages of 8 chunks
    uint32_t dbl1, dbl2;    // stages before which the accumulators double (weight classes), PO_NO_DOUBLING = never
    uint64_t op_n;          // records per chunk row of the operand (n rounded up to the 256-record tile)
    uint64_t k_elems;       // K (KT: padded; BC: thermometer planes before padding)
    size_t aux_offset;      // BC: where the per-record count sums sit inside ws_thermo
};
bool po_kt_pairdot_supported(uint32_t dim);
size_t po_kt_pairdot_rank_bytes(uint64_t n, uint32_t dim);
size_t po_kt_pairdot_operand_bytes(uint64_t n, uint32_t dim, uint32_t words, uint32_t n_selfs, bool folded, int fmt_fp4);
// The materialised operand lives in the context's workspace until a call for another metric, po_ctx_trim or po_ctx_destroy.
// 96 GB of the 288 GB of HBM3E (round 4; 24 GB before): Kendall at k = 6 is 1.05 MB of FP4 pair signs per record, so the matrix-core
// path now reaches ~90 000 records there (3.4 x the panel kernel, tools/exp/kt_big.py); a failed allocation falls back to the panel kernel
#define PO_PAIRDOT_MAX_OPERAND (96ull << 30)
int po_launch_kt_pairdot_prep(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, uint64_t npad,
                              const uint32_t* fold_src, uint32_t n_selfs, uint32_t n_pairs, int fmt_fp4,
                              po_pairdot_plan* plan);
int po_launch_kt_pairdot_tiles(po_ctx* ctx, const po_tile_args& a, const po_pairdot_plan& plan, uint64_t* tiles);
// Bray-Curtis on thermometer planes; p8t / groups_pad / cls from the SAD prep (po_bc_sad_view)
void po_bc_sad_view(const void* ws, uint64_t npad, uint32_t dim, const uint32_t** p8t, uint32_t* groups_pad);
int po_launch_bc_thermo_prep(po_ctx* ctx, const uint32_t* p8t, uint32_t groups_pad, const unsigned long long* cls, uint64_t n,
                             uint32_t dim, uint64_t npad, uint32_t dbl_at, int fmt_fp4, bool* eligible, po_pairdot_plan* plan,
                             double* inv_n);
int po_launch_bc_thermo_tiles(po_ctx* ctx, const po_tile_args& a, const po_pairdot_plan& plan, double inv_n, uint64_t* tiles);

// Kendall on the int8 matrix cores, 256 < dim <= 16384: 64-word rank panels (po_kt_panel.hip)
bool po_kt_panel_supported(uint32_t dim);
bool po_kt_panel_fold_supported(uint32_t dim, uint32_t n_selfs);
size_t po_kt_panel_workspace(uint64_t n, uint32_t dim);
struct po_kt_panel_plan {
    uint32_t n_diag_items, words, row_words, self_panels;
    int folded;
};
int po_launch_kt_panel_prep(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, uint64_t npad, void* ws,
                            const uint32_t* fold_src, uint32_t fold_src_len, uint32_t n_selfs, uint32_t n_pairs,
                            po_kt_panel_plan* plan);
int po_launch_kt_panel_tiles(po_ctx* ctx, const po_tile_args& a, const void* ws, const po_kt_panel_plan& plan, uint64_t* tiles);

static inline uint64_t po_round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }
