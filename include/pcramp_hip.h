/*
 * pcramp_hip.h -- C-ABI of the MI355X-native primer-pair x target evaluation path.
 *
 * The reference (LANL-Bioinformatics/PCRamp v0.3) has no plugin/FFI layer: its boundary for
 * this path is a set of C++ member functions called from main.cpp / optimize.cpp.  Each entry
 * point below replaces the reference call sites it cites (file:line under the reference tree)
 * and is what a `pcramp` host driver binds instead of them.
 *
 * Conventions: C linkage, opaque handle, plain pointers + sizes, caller-owned buffers, `int`
 * status (0 = ok, <0 = error; text via pcr_last_error(), thread-local).  No exceptions cross
 * the ABI.  A handle owns one GPU and one HIP stream and is not thread-safe (the reference's
 * equivalent rule: one NucCruc / SeqOverlap per OpenMP thread, main.cpp:533,702).
 * There is NO CPU fallback: every compute entry point fails if no gfx950 device is usable.
 */
#ifndef PCRAMP_HIP_H
#define PCRAMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_OK               0
#define PCR_ERR_ARG         -1
#define PCR_ERR_DEVICE      -2   /* no usable GPU / HIP failure */
#define PCR_ERR_STATE       -3   /* call order (e.g. amplify before select) */
#define PCR_ERR_CAPACITY    -4   /* a fixed device buffer would overflow even after growth */
#define PCR_ERR_RANGE       -5   /* reference `throw` conditions (e.g. Sequence::has_split out of bounds) */

typedef struct pcr_ctx pcr_ctx;

/* Reference `Word` (= __word<unsigned long,2>, word.h:690): 32 slots of 4-bit IUPAC codes,
 * slot k (0 = 5' end) at bits (15 - k%16)*4 of w[k/16]; A=1 C=2 G=4 T=8, EOS=0 (base_table.h:11-28). */
typedef struct { uint64_t w[2]; } pcr_word128;

/* One trial assay: PCR::f / PCR::r (assay.h:117-118). */
typedef struct { pcr_word128 f, r; } pcr_pair;

/* The sequence sets of a context: target_seq, background_seq (main.cpp:257-344) and multiplex_background_seq, the amplicons of
 * the assays accepted so far as sequences (main.cpp:989-1001; templates of find_multiplex_background_match). */
typedef enum { PCR_SET_TARGET = 0, PCR_SET_BACKGROUND = 1, PCR_SET_MULTIPLEX = 2 } pcr_set;
#define PCR_N_SETS 4   /* + one internal scratch set */

/* The `Options` fields (pcramp.h:83-128) that Sequence::pack reads (sequence.cpp:92-96). */
typedef struct {
	uint32_t pack_max_degen;  /* opt.pack_max_degen, default 256 (pcramp.h:43) */
	float pack_min_gc;        /* opt.pack_min_gc, default 0 = off (pcramp.h:45) */
	float pack_max_gc;        /* opt.pack_max_gc, default 1 = off (pcramp.h:44) */
} pcr_params;

/* One entry of the per-iteration word DB (`MULTIMAP<Word, WordMatch>`, sequence.h:34-76). */
typedef struct {
	pcr_word128 word;
	int32_t loc;       /* WordMatch::loc */
	uint32_t index;    /* WordMatch::index (sequence index inside its set) */
	uint32_t strand;   /* 1 = Seq_strand_plus, 2 = Seq_strand_minus (sequence.h:27-32) */
	uint32_t pad;
} pcr_entry;

/* Arguments of the amplicon screen; mirrors what PCR::collect_candidates (pcr_assay.cpp:12-18)
 * and PCR::find_target_match / compute_coverage (pcr_assay.cpp:544,271) take from `Options`. */
typedef struct {
	float collect_threshold;  /* m_threshold of collect_candidates: target_threshold (find_target_match,
	                             pcr_assay.cpp:556) or target_threshold*search_multiplier (assay.h:407);
	                             squared inside, as pcr_assay.cpp:31-32 */
	float ident_threshold;    /* the sqrtf(f*r) >= threshold test (pcr_assay.cpp:574, :294) */
	int32_t amp_min, amp_max; /* m_amplicon_range (pcramp.h:14-18) */
	int32_t use_taq_mama;     /* opt.use_taq_mama (optimize.cpp:209) */
} pcr_amplify_args;

const char *pcr_last_error(void);

/* Device index + an optional caller HIP stream (hipStream_t as void*, NULL = create one).
 * Fails (returns NULL) when no gfx950 device is available. */
pcr_ctx *pcr_create(int device, void *hip_stream, const pcr_params *params);
void pcr_destroy(pcr_ctx *ctx);

/* Replaces the host-resident `deque<Sequence>` (main.cpp:257-344 after parse_fasta):
 * `packed4` holds the sequences' 4-bit codes, high nibble first (sequence.h:223-228);
 * sequence i starts at byte byte_offsets[i] and has lengths[i] bases; weights[i] = Sequence::weight().
 * Builds the HBM-resident bit-plane store, the window-validity mask (the degeneracy / GC /
 * EOS filters of Sequence::pack, sequence.cpp:127-153) and the irregular word list (centred
 * partial words at sequence ends and around EOS, sequence.cpp:155-179,198-263). */
int pcr_load_sequences(pcr_ctx *ctx, pcr_set which, const uint8_t *packed4,
	const uint64_t *byte_offsets, const uint64_t *lengths, const float *weights, uint32_t n);

/* Sequence::active(bool) for every sequence of the set (main.cpp:1105-1120); active[i] != 0 = active. */
int pcr_set_active(pcr_ctx *ctx, pcr_set which, const uint8_t *active);

/* Sequence::split_sequence (sequence.h:228-241; main.cpp:1008-1017): write EOS at `pos`. */
int pcr_split(pcr_ctx *ctx, pcr_set which, uint32_t seq, uint64_t pos);
/* n splits at once (the three per amplicon of an accepted assay, main.cpp:1008-1017: a few hundred per design iteration): the
 * derived device state -- window validity around each split, the set's irregular word list -- is refreshed once. */
int pcr_split_many(pcr_ctx *ctx, pcr_set which, const uint32_t *seq, const uint64_t *pos, uint32_t n);

/* The per-iteration index build, main.cpp:579-615 / 644-691: for every ACTIVE sequence
 * Sequence::pack (sequence.cpp:92) + select_words (select_words.cpp:8) with the trial assays
 * `pairs`; the result (the set's word DB) stays on the device.  `threshold` is the
 * select_words m_threshold (target_threshold*target_search_multiplier, main.cpp:669; not squared);
 * `min_oligo_length` is the pack argument (opt.min_oligo_length(), x0.9 for backgrounds, main.cpp:595).
 * n_entries_out (optional) receives the DB size. */
int pcr_select_words(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	int optimize_5, int optimize_3, float threshold, uint32_t min_oligo_length,
	uint64_t *n_entries_out);

/* Copy the current word DB to the host (parity tests; the reference's target_db). Returns the
 * number of entries (may exceed cap; only cap are written) or <0. */
int64_t pcr_get_entries(pcr_ctx *ctx, pcr_set which, pcr_entry *out, uint64_t cap);

/* The amplicon screen for a batch of pairs against every sequence of the set:
 * PCR::collect_candidates + update_identity + the sqrtf(f*r) test, i.e. PCR::find_target_match
 * (pcr_assay.cpp:544-578) and PCR::compute_coverage (pcr_assay.cpp:271-302) in one pass.
 *   bits      : n_pairs x pcr_bitset_words(ctx,which) u64 words, bit (i%64) of word i/64 = sequence i
 *               amplified by the pair (BitSet, bitset.h:7).  May be NULL.
 *   bits_fr   : same shape, amplified through {F(+),R(-)} (pcr_assay.cpp:39-45).  May be NULL.
 *   bits_rf   : same shape, amplified through {R(+),F(-)} (pcr_assay.cpp:52-58).  May be NULL.
 *   coverage  : n_pairs floats = compute_coverage's return value (double sum in the reference's
 *               amplicon order, then rounded to float).  May be NULL. */
int pcr_amplify(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	const pcr_amplify_args *args, uint64_t *bits, uint64_t *bits_fr, uint64_t *bits_rf,
	float *coverage);

/* Same screen, asynchronous on the handle's stream, results left in DEVICE memory
 * (d_bits_fr / d_bits_rf: n_pairs x words each, caller-allocated, e.g. a torch tensor that is
 * then all-gathered over RCCL).  No host synchronisation. */
int pcr_amplify_device(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	const pcr_amplify_args *args, uint64_t *d_bits_fr, uint64_t *d_bits_rf);

/* pcr_select_words followed by pcr_amplify_device for the same batch -- one optimiser iteration's
 * DB build + find_target_match (main.cpp:644-691 then pcr_assay.cpp:544-578) -- ENQUEUED on the
 * handle's stream without any host wait, so the host plans pass i+1 while pass i runs.
 * The per-sequence bucket overflow that pcr_select_words handles by retrying cannot be seen
 * before the pass has run: the device buffers are final only after pcr_synchronize() (or any other
 * entry point of the same handle) has returned PCR_OK -- that call inspects the counters of every
 * enqueued pass and, if a pass overflowed, replays it and the later ones with larger buckets into
 * the same buffers.  At most 6 passes are kept in flight. */
int pcr_screen_device(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	int optimize_5, int optimize_3, float select_threshold, uint32_t min_oligo_length,
	const pcr_amplify_args *args, uint64_t *d_bits_fr, uint64_t *d_bits_rf);

/* One local-search move, evaluated as optimize_pcr.cpp does for each of its six moves (e.g.
 * increase_degeneracy :57-100; the loop of optimize(), optimize.cpp:120-140, calls them per oligo):
 * the candidate amplicons are those of the BASE pair -- collected at args->collect_threshold
 * (= target_threshold*search_multiplier, optimize.cpp:61-63, assay.h:405-408) against the word DB of
 * the last pcr_select_words -- and for every variant of the edited oligo (side 0 = F, 1 = R) only that
 * oligo's identity table is recomputed (update_identity, optimize.cpp:209-261: the variant's own
 * length and 3' bases) before compute_coverage at args->ident_threshold (pcr_assay.cpp:271-302).
 *   variants  : n_variants trial words (as the move leaves them, not re-centred)
 *   bits_fr / bits_rf : n_variants x pcr_bitset_words u64, per orientation as pcr_amplify.  May be NULL.
 *   coverage  : n_variants floats = compute_target_coverage of each trial.  May be NULL. */
int pcr_move_coverage(pcr_ctx *ctx, pcr_set which, const pcr_pair *base, int side,
	const pcr_word128 *variants, uint32_t n_variants, const pcr_amplify_args *args,
	uint64_t *bits_fr, uint64_t *bits_rf, float *coverage);

/* compute_coverage's weight sum (pcr_assay.cpp:280-301) from gathered orientation bitsets:
 * ascending index over bits_fr, then ascending index over bits_rf & ~bits_fr, accumulated in
 * double; n = number of sequences, weights[n].  Pure host arithmetic (no device needed). */
float pcr_coverage_from_bits(const uint64_t *bits_fr, const uint64_t *bits_rf,
	const float *weights, uint64_t n);

/* weighted_coverage (main.cpp:1402-1418): ascending-index double sum of weights over set bits. */
float pcr_weighted_coverage(const uint64_t *bits, const float *weights, uint64_t n);

uint32_t pcr_num_sequences(pcr_ctx *ctx, pcr_set which);
uint64_t pcr_bitset_words(pcr_ctx *ctx, pcr_set which);   /* ceil(n/64) */

/* Measurement hooks (bench.py): when enabled, the dominant kernel of pcr_select_words (the
 * oligo x window match scan) is bracketed by HIP events on the handle's stream.  on = 1: every pass;
 * on = n > 1: every n-th pass (an event packet between two kernels costs a ~6 us queue bubble).
 * pcr_profile_read: total milliseconds and number of bracketed launches since the last reset. */
int pcr_profile_enable(pcr_ctx *ctx, int on);
int pcr_profile_read(pcr_ctx *ctx, double *scan_ms, uint64_t *scan_launches, int reset);
/* The same for the other kernels bench.py prices: while profiling is on, EVERY launch of the Smith-Waterman kernel
 * (k_sw) and of the thermodynamics kernels is bracketed by HIP events on the handle's stream. */
#define PCR_PROF_SCAN    0   /* = pcr_profile_read */
#define PCR_PROF_SW      1
#define PCR_PROF_THERMO  2
#define PCR_PROF_KERNELS 3
int pcr_profile_read_kernel(pcr_ctx *ctx, int kernel, double *ms, uint64_t *launches, int reset);

/* Blocks until the handle's stream is idle. */
int pcr_synchronize(pcr_ctx *ctx);

/* How the small per-pass tables of pcr_screen_device reach the device (reported by bench.py): 1 = "lean" -- the CPU
 * stores them straight into fine-grained device memory (large BAR; probed by pcr_create with a store / fence / kernel
 * read-back round trip) and the pass has no staging launch; 0 = a staging kernel (k_stage) copies them out of mapped
 * host memory (PCRAMP_STAGE=kernel, no large BAR, or a failed probe). */
int pcr_staging_mode(pcr_ctx *ctx);

/* ---- multi-GPU: the path's one exchange step.  Targets shard across ranks (one process per GPU,
 * contiguous blocks whose boundaries are multiples of 64 sequences, so every rank owns whole bitset words); the primer pairs are
 * replicated; after a pass every rank needs every rank's orientation bitsets.  This replaces what the reference's MPI mode does
 * with its per-rank best assay -- Send/Recv of the BitSets to rank 0 and MPI_Bcast of the winner (main.cpp:1421-1601; wire
 * format mpi_util.cpp:152-233) -- by ONE ncclAllGather over xGMI, enqueued on the handle's stream (no host hop, no torch).
 *   pcr_comm_unique_id   rank 0: 128 opaque bytes (RCCL's ncclUniqueId) to hand to every rank (the reference's MPI_Bcast, a file, ...)
 *   pcr_comm_init_rank   every rank, once: joins the communicator with this handle's device (any handle on that device may then
 *                        exchange through it); NULL + pcr_last_error() on failure
 *   pcr_exchange_bits    every rank, per pass: all-gather of words_per_rank u64 from d_local (device) into d_full (device,
 *                        world x words_per_rank, rank-major).  Waits on the host only for the counters of passes pcr_screen_device
 *                        enqueued (a bucket overflow replays a pass into the same buffers), then enqueues the collective; the
 *                        result is complete after pcr_synchronize / in stream order for later kernels on the handle's stream.
 * Coverage is then pcr_coverage_from_bits on every rank over the gathered words, in the reference's summation order.
 * RCCL is bound at run time (the copy already loaded in the process first; PCRAMP_RCCL=<path> overrides). */
#define PCR_COMM_ID_BYTES 128
typedef struct pcr_comm pcr_comm;
int pcr_comm_unique_id(uint8_t id[PCR_COMM_ID_BYTES]);
pcr_comm *pcr_comm_init_rank(pcr_ctx *ctx, const uint8_t id[PCR_COMM_ID_BYTES], int world, int rank);
int pcr_comm_world(const pcr_comm *comm);
int pcr_comm_rank(const pcr_comm *comm);
int pcr_exchange_bits(pcr_ctx *ctx, pcr_comm *comm, const uint64_t *d_local, uint64_t words_per_rank, uint64_t *d_full);
void pcr_comm_destroy(pcr_comm *comm);
const char *pcr_comm_library(void);      /* which librccl was bound ("" before the first use / when none was found) */


/* ---- Smith-Waterman primer x template alignment (rows a7/a8 of the scope table) */

/* Outputs of one SO::SeqOverlap lane (seq_overlap.h:1266-1330). */
typedef struct {
	int16_t score;              /* SeqOverlap::score() = max_elem.M */
	int16_t q_start, q_stop;    /* alignment_range_query() */
	int16_t t_start, t_stop;    /* alignment_range_target() */
	uint8_t last1, last2;       /* target_last_two_aligned() (15,15 = N,N when undefined) */
	uint8_t valid;              /* 0: no cell reached the running maximum; the reference leaves the
	                               coordinates stale then, here they are zero and only score (0) is defined */
	uint8_t pad;
} pcr_sw_result;

/* n independent alignments, query word i against template word i, exactly as
 * SeqOverlap::pack_query_slots / pack_target_slots(Word) + align() (SmithWaterman, nucleic acid)
 * evaluate one lane (seq_overlap.h:828,1099; seq_overlap.cpp:347-609). */
int pcr_sw_align_words(pcr_ctx *ctx, const pcr_word128 *queries, const pcr_word128 *templates, uint32_t n,
	pcr_sw_result *out);

typedef struct {
	float collect_threshold;     /* background_threshold*background_search_multiplier (assay.h:418); squared inside */
	float background_threshold;  /* the final score test (background_match.cpp:116) */
	int32_t amp_min, amp_max;    /* opt.background_amplicon_range */
	int32_t use_taq_mama;
	int32_t evaluate_all_amplicons;  /* 0 (default): reference-identical -- see below; 1: every candidate amplicon counts */
} pcr_background_args;

/* PCR::find_background_match (background_match.cpp:7-166) for a batch of pairs against the set's
 * current word DB: candidate amplicons -> 4 alignments each -> normalised score product ->
 * bits[pair][seq].
 * The reference aligns its candidate amplicons two per SeqOverlap call (:66-75) and tests
 * `(i + 1) >= num_seq` (:122, the number of SEQUENCES, not of amplicons) before scoring the second one:
 * in the order collect_candidates leaves them (pcr_assay.cpp:39-58: {F(+),R(-)} first, then {R(+),F(-)},
 * each by sequence, plus site, minus site) an amplicon with an odd index >= the set's sequence count is
 * never scored.  With evaluate_all_amplicons = 0 that is reproduced bit for bit; with 1 every candidate
 * amplicon is scored.  (One case stays ours in both modes: an odd amplicon COUNT below num_seq makes the
 * reference score stale SSE lanes and index past its deque -- undefined there, no extra amplicon here.) */
int pcr_background_match(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	const pcr_background_args *args, uint64_t *bits);

/* PCR::find_multiplex_background_match (background_match.cpp:168-295): F, (F), R, (R) of every
 * pair aligned against every whole sequence of the set (amplicon sequences, up to 32767 bases);
 * bit set when any single normalised score reaches the threshold. */
int pcr_multiplex_match(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs,
	float background_threshold, int use_taq_mama, uint64_t *bits);

/* ---- Nearest-neighbour thermodynamics (rows a9/a10 of the scope table) */

/* The `Options` fields the thermodynamic filters read (pcramp.h:21-30). */
typedef struct {
	float salt;            /* opt.salt -> NucCruc::salt() (main.cpp:535) */
	float primer_strand;   /* opt.primer_strand */
	float tm_min, tm_max;  /* opt.primer_tm_range */
	float max_hairpin;     /* opt.max_hairpin */
	float max_dimer;       /* opt.max_dimer */
} pcr_thermo_args;

typedef struct {
	uint32_t valid;        /* PCR::is_valid(): every IUPAC expansion passes all enabled tests */
	uint32_t n_expansions; /* Word::degeneracy() as an integer */
	float tm, dH, dS;      /* NucCruc::tm_pm_duplex / delta_H / delta_S of the FIRST expansion (Word::begin()); kcal/mol, kcal/(mol K) */
	float hairpin_tm;      /* approximate_tm_hairpin of the first expansion */
	float homodimer_tm;    /* approximate_tm_homodimer of the first expansion (0 unless check_homo_dimer) */
	float dG;              /* NucCruc::delta_G() = dH - 310.15 K * dS of the perfect-match duplex (nuc_cruc.h:1375-1378), first expansion */
} pcr_thermo_result;

/* PCR::is_valid (valid_pcr.cpp:5-45) for n oligos: perfect-match duplex Tm in [tm_min, tm_max],
 * hairpin Tm <= max_hairpin and, if check_homo_dimer, homodimer Tm <= max_dimer, for every
 * non-degenerate expansion, at strand concentration primer_strand/degeneracy (valid_pcr.cpp:13). */
int pcr_thermo(pcr_ctx *ctx, const pcr_word128 *oligos, uint32_t n, int check_homo_dimer,
	const pcr_thermo_args *args, pcr_thermo_result *out);

/* PCR::max_dimer_tm (pcr_assay.cpp:232-269): max heterodimer Tm of F x R over all expansions,
 * strand(primer_strand/D(F), primer_strand/D(R)) (nuc_cruc.h:818-838). */
int pcr_dimer(pcr_ctx *ctx, const pcr_pair *pairs, uint32_t n, const pcr_thermo_args *args, float *max_tm);

/* PCR::multiplex_compatible (pcr_assay.cpp:815-852): ok[i] = 1 iff no oligo expansion of assay
 * a[i] forms a heterodimer with any oligo expansion of assay b[i] at Tm >= max_dimer
 * (strand = primer_strand, no degeneracy correction). */
int pcr_multiplex_compatible(pcr_ctx *ctx, const pcr_pair *a, const pcr_pair *b, uint32_t n,
	const pcr_thermo_args *args, uint8_t *ok);

/* ---- Multiplex background (the third coverage term of the multiplex local search) */

/* The multiplex background DB and its keys (main.cpp:989-1001): every word Sequence::pack emits for the
 * amplicon sequences of the assays accepted so far (pack_max_degen of pcr_create, no G+C filter,
 * min_oligo_length = opt.min_oligo_length()), made unique.  The call REPLACES the key table: pass all amplicons
 * collected so far.  Sequence format as pcr_load_sequences.  n_keys_out (optional) = keys(db).size(). */
int pcr_multiplex_load(pcr_ctx *ctx, const uint8_t *packed4, const uint64_t *byte_offsets, const uint64_t *lengths,
	uint32_t n, uint32_t min_oligo_length, uint64_t *n_keys_out);

/* collect_multiplex_background_candidates (pcr_assay.cpp:71-102) for `base`, then, for every trial word of the
 * oligo on `side` (0 = F, 1 = R), update_identity of that oligo's map (optimize.cpp:209-261) and
 * compute_multiplex_background_coverage (pcr_assay.cpp:304-336): coverage[v] = number of distinct candidate
 * keys whose identity with the forward or with the reverse oligo is >= background_threshold.  The unedited
 * assay is the variant `base->f` (side 0). */
int pcr_multiplex_coverage(pcr_ctx *ctx, const pcr_pair *base, int side, const pcr_word128 *variants, uint32_t n_variants,
	float background_threshold, int use_taq_mama, float *coverage);

/* One amplicon an assay produces on a sequence of the set (AmpliconBounds, assay.h:64-89, plus the stretch
 * extract_amplicon_seq spells for the multiplex background, pcr_assay.cpp:489-497). */
typedef struct {
	uint32_t sequence;               /* AmpliconBounds::index */
	int32_t  begin, end;             /* first / last base of the amplicon INCLUDING both primers */
	int32_t  inner_start;            /* first base of the stretch that becomes a multiplex background sequence ... */
	int32_t  inner_length;           /* ... and its length (non-primer part + MULTIPLEX_AMPLICON_PADDING) */
	uint32_t orientation;            /* 0: F on the plus strand, 1: R on the plus strand */
} pcr_amplicon;

/* PCR::collect_unique_amplicons (pcr_assay.cpp:756-813; main.cpp:787,920) for one assay over the word DB of the
 * last pcr_select_words on `which`: every (plus site, minus site) pair of an active sequence whose oligos match
 * at threshold^2, do not overlap, span amp_min..amp_max bases and enclose no EOS.  Records sorted by
 * (orientation, sequence, begin, end).  Returns the number found (may exceed cap; only cap are written) or a
 * negative error.  The caller cuts the strings from its Sequences, makes them unique (sort + unique, :805-806),
 * passes them to pcr_multiplex_load and splits the targets at begin, (begin+end)/2 and end (main.cpp:1008-1017). */
int64_t pcr_collect_amplicons(pcr_ctx *ctx, pcr_set which, const pcr_pair *pair, float threshold, int32_t amp_min, int32_t amp_max,
	pcr_amplicon *out, uint64_t cap);

/* ---- The multiplex compatibility filter of the trial loop (main.cpp:744-803), batched over the trial assays */

typedef struct {
	pcr_thermo_args thermo;        /* salt, primer_strand, max_dimer: PCR::multiplex_compatible (pcr_assay.cpp:815-852) */
	float background_threshold;    /* opt.background_threshold (find_multiplex_background_match) */
	int32_t use_taq_mama;
	float target_threshold;        /* opt.target_threshold, amplicon range: collect_unique_amplicons (main.cpp:787-789) */
	int32_t amp_min, amp_max;
} pcr_multiplex_screen_args;

/* For every trial assay t, what main.cpp:744-803 computes against the pool of accepted assays:
 *   compatible[t]      = every pool assay is multiplex_compatible with it (:748-752);
 *   multiplex_cover[t] = weighted_coverage of trial[t].find_multiplex_background_match over PCR_SET_MULTIPLEX (:767-771);
 *   pool_cover[t]      = weighted_coverage of the union, over the pool assays, of find_multiplex_background_match against the
 *                        trial's own unique amplicons (collect_unique_amplicons on the targets at target_threshold; amplicon
 *                        Sequences carry the default weight 1, so this is the number of unique amplicons some pool assay
 *                        matches; :786-803).
 * The two covers are computed for the trials with detail[t] != 0 (NULL: all) that are compatible; others get 0.  The reference
 * gates them on best_score < s and on max_background_cover as it walks the trials (:754, :778): the caller applies those to
 * the returned numbers.  One thermodynamics launch for all (trial, pool) pairs, one alignment pass for all trials over the
 * multiplex set, one for all pool assays over all trials' amplicons (loaded into an internal scratch set).  Needs the target
 * word DB selected for the trial batch (as pcr_collect_amplicons does). */
int pcr_multiplex_screen(pcr_ctx *ctx, const pcr_pair *trial, uint32_t n_trial, const pcr_pair *pool, uint32_t n_pool,
	const pcr_multiplex_screen_args *args, const uint8_t *detail, uint8_t *compatible, float *multiplex_cover, float *pool_cover);

/* ---- Random assay sampler (scope row f-2) */

/* The `Options` fields PCR::random_assay reads besides the thermodynamic ones (pcramp.h:21-30). */
typedef struct {
	int32_t primer_min, primer_max;  /* opt.primer_range */
	int32_t amp_min, amp_max;        /* opt.target_amplicon_range */
	double  max_degen;               /* opt.degen */
} pcr_sampler_args;

typedef struct {
	uint32_t sequence;               /* index of the sequence the assay was cut from */
	int32_t  f_start;                /* 0-based start of the forward primer */
	int32_t  amplicon_length;        /* the "Amplicon length" of the verbose line, pcr_assay.cpp:728 */
	uint32_t sequence_iterations;    /* "tried N seq ..." */
	uint32_t assay_iterations;       /* "... and M assays" (of the last sequence tried) */
} pcr_sample_info;

/* PCR::random_assay (pcr_assay.cpp:580-734) for n_trials fresh assays drawn one after the other from the
 * running glibc rand_r state *seed -- the body of the sampling loop main.cpp:544-550 as one thread executes
 * it (the caller obtains that thread's seed with pcr_host_rand_r(&global_seed), main.cpp:541-542).
 * Sequences, their active flags and their EOS splits are those of set `which` as loaded.  Validity
 * (is_valid with the homodimer test, max_dimer_tm) is evaluated on the device: the thermodynamic jobs of
 * every attempt that could start inside a window of the random stream go out as one launch.
 * pairs_out[i] = the centred assay of trial i; info_out is optional.  Where the reference throws
 * ("No active sequences found", "sequence length is too small!", "Unable to generate a valid initial
 * assay to test!", Sequence::subword out of bounds) an error code is returned with that text. */
int pcr_random_assays(pcr_ctx *ctx, pcr_set which, uint32_t *seed, uint32_t n_trials, const pcr_sampler_args *sampler,
	const pcr_thermo_args *thermo, pcr_pair *pairs_out, pcr_sample_info *info_out);

/* Word::max_overlap (word.h:38-91): the best ungapped diagonal of equal codes between two oligo words, as a
 * fraction of the longer one -- the oligo-reuse term of the multiplex Score (pcramp.h:158-208).  Host arithmetic. */
float pcr_host_max_overlap(const pcr_word128 *a, const pcr_word128 *b);

/* out[i] = max over both oligos of every pooled assay of Word::max_overlap(words[i], .) -- the inner loops of the
 * moves' oligo-reuse term (optimize_pcr.cpp:27-53, :129-139) for a batch of trial words; 0 for an empty pool. */
int pcr_host_pool_overlaps(const pcr_word128 *words, uint32_t n, const pcr_pair *pool, uint32_t n_pool, float *out);

/* PCR::compute_oligo_overlap (pcr_assay.cpp:736-754): both oligos of `assay` against both oligos of every pooled
 * assay, with MULTIPLEX_OLIGO_REUSE_BONUS (assay.h:19) for an identical oligo.  Host arithmetic. */
float pcr_host_oligo_overlap(const pcr_pair *assay, const pcr_pair *pool, uint32_t n_pool);

/* glibc rand_r (the reference's random source, sample.cpp:12), restated; usable without a GPU. */
uint32_t pcr_host_rand_r(uint32_t *seed);

/* ---- The local search behind the ABI, batched over trial assays (scope row f-1) */

/* The `Options` fields optimize() and its moves read (pcramp.h:83-128). */
typedef struct {
	double  max_degen;               /* opt.degen */
	int32_t primer_min, primer_max;  /* opt.primer_range */
	pcr_thermo_args thermo;          /* is_valid of the trial oligos: salt, primer_strand, tm range, max_hairpin (no dimer test there) */
	pcr_amplify_args target;         /* collect_threshold = target_threshold*target_search_multiplier, ident_threshold = target_threshold,
	                                    target amplicon range, use_taq_mama (optimize.cpp:61-77) */
	pcr_amplify_args background;     /* the same with the background thresholds and amplicon range */
	int32_t have_background;         /* 0: no background sequences, background coverage is 0 */
	int32_t use_multiplex;           /* opt.use_multiplex: multiplex background coverage (pcr_multiplex_load) and oligo reuse join the Score */
	float   multiplex_threshold;     /* opt.background_threshold, the threshold of the multiplex background coverage */
	int32_t n_moves;                 /* optimization_moves (main.cpp:82-95), at most 8, in order: PCR_MOVE_* */
	int32_t moves[8];
} pcr_optimize_args;

/* optimize() (optimize.cpp:14-207) for n assays at once -- the body of the parallel loop main.cpp:697-887 runs for every
 * trial assay of a design iteration.  The word DBs of the target set (and of the background set if have_background) must
 * have been built for these assays (pcr_select_words), as main.cpp builds them before it optimises.  All assays advance
 * in lockstep; per optimiser iteration ONE thermodynamics launch decides is_valid for every trial word of every assay
 * and ONE move-coverage pass per sequence set evaluates them (see pcr_optimize.inc).  pool = the assays designed so
 * far (use_multiplex).
 *   best_out[i]      : the best assay found for assays[i] (oligos centred as optimize.cpp:152 leaves them)
 *   score_out[3*i..] : its Score {target_coverage, background_coverage, oligo_overlap} (pcramp.h:158-208); optional
 *   iterations_out[i]: optimiser iterations it took; optional */
int pcr_optimize_batch(pcr_ctx *ctx, const pcr_pair *assays, uint32_t n, const pcr_optimize_args *args, const pcr_pair *pool, uint32_t n_pool,
	pcr_pair *best_out, float *score_out, uint32_t *iterations_out);

/* make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy, pcr_assay.cpp:111-230) for n trial assays: the top-down start of
 * the local search (--optimize.top-down; main.cpp:709-721 calls it before optimize() and drops the trial when it returns false).
 * The target word DB must have been built for these assays (pcr_select_words).  Reads of args: max_degen (opt.degen), thermo (salt,
 * primer_strand, Tm range, max_hairpin, max_dimer: PCR::is_valid with the homodimer test, PCR::max_dimer_tm), target (collect
 * threshold, amplicon range, use_taq_mama).  assays[i] is replaced by the maximally degenerate assay; valid[i] = what the reference
 * returns (0: the greedy reduction ended non-degenerate with a heterodimer still above max_dimer).  Ties in the reference's
 * std::sort of the candidate amplicons by score are resolved by the same libstdc++ algorithm on the same initial order. */
int pcr_make_degenerate(pcr_ctx *ctx, pcr_pair *assays, uint32_t n, const pcr_optimize_args *args, uint8_t *valid);

/* optimization_move (optimize.cpp:303-352): ONE move (PCR_MOVE_*) of ONE oligo (side 0 = F, 1 = R) of one assay.
 * score_threshold (3 floats, optional): the Score trials are measured against; NULL = the unmodified assay's own Score,
 * as optimization_move computes it.  word_out = the winning trial word (not re-centred; all zero if no trial survives),
 * score_out = its Score (Score() = {-1e6, 1e6, 0} then), base_score_out (optional) = the unmodified assay's Score. */
int pcr_optimization_move(pcr_ctx *ctx, const pcr_pair *assay, int move, int side, const pcr_optimize_args *args, const pcr_pair *pool, uint32_t n_pool,
	const float *score_threshold, pcr_word128 *word_out, float *score_out, float *base_score_out);

/* ---- Assay-list writers (scope row f-4): the bytes `pcramp` writes to its output file.  Host only. */

/* What the writers read besides the assay itself (Options::output_format, opt.use_multiplex, the sequence records). */
typedef struct {
	int32_t json;                        /* 0 = Options::TEXT_OUTPUT, 1 = Options::JSON_OUTPUT (pcramp.h, --o.text / --o.json) */
	int32_t use_multiplex;               /* opt.use_multiplex (always true in v0.3, options.cpp:72): reused oligos are marked */
	uint32_t n_target, n_background;
	const char *const *target_deflines;       /* Sequence::defline(); keeps the leading '>' (parse_fasta.cpp:58) */
	const char *const *background_deflines;
	const uint64_t *target_lengths;           /* Sequence::length(), for sequence_summary (main.cpp:1326-1400) */
	const uint64_t *background_lengths;
} pcr_output;

/* One accepted assay as main.cpp:950-1113 writes it. */
typedef struct {
	uint32_t major_id, minor_id;              /* main.cpp:468-502 */
	pcr_pair assay;                           /* best_assay */
	float target_coverage, background_coverage;        /* best_score */
	float active_target_norm, active_background_norm;  /* summed weights of the active sequences (main.cpp:603,649) */
	uint32_t num_active_background;           /* main.cpp:565-569 */
	const uint64_t *target_match;             /* best_target_match: BitSet words, bit i%64 of word i/64 */
	const uint64_t *background_match;         /* best_background_match (may be NULL when there are no backgrounds) */
} pcr_assay_record;

/* All writers return the number of bytes of the text (the NUL is not counted) or a negative error; at most `cap`
 * bytes are stored (NUL-terminated when they fit), so a first call with cap = 0 sizes the buffer. */

/* PCR::write(ostream&) / write(ostream&, pool) / write_json(ostream&) / write_json(ostream&, pool), assay.h:288-375:
 * "F\tR\tD(F)=..;D(R)=.." with reused oligos in lower case (text), or the two primer objects with "recycled":True|False
 * (JSON).  use_multiplex = 0 selects the pool-less forms. */
int64_t pcr_format_oligos(const pcr_pair *assay, const pcr_pair *pool, uint32_t n_pool, int json, int use_multiplex,
	char *out, uint64_t cap);

/* Version, command line and seed (main.cpp:131-163), both sequence summaries (main.cpp:440-443) and, for JSON, the
 * opening of the assay array (main.cpp:460-462). */
int64_t pcr_format_header(const pcr_output *o, int argc, const char *const *argv, uint32_t seed, char *out, uint64_t cap);

/* What every design iteration writes BEFORE the search (main.cpp:504-519): the rule and "# Attempting to detect N
 * remaining targets" (text) or the record opening with its id (JSON).  The reference writes it even if the iteration
 * then finds no assay. */
int64_t pcr_format_iteration(const pcr_output *o, uint32_t assay_iteration, uint32_t major_id, uint32_t minor_id,
	uint32_t targets_remaining, char *out, uint64_t cap);

/* One accepted assay (main.cpp:950-1113): score line + "ASSAY.M.m" + oligos + "T-"/"B-" deflines (text), or the primer
 * objects + "target matches" + "background matches" (JSON).  pool = the assays accepted before this one. */
int64_t pcr_format_assay(const pcr_output *o, const pcr_assay_record *rec, const pcr_pair *pool, uint32_t n_pool,
	char *out, uint64_t cap);

/* The end of the file (main.cpp:1132-1264): targets still active (= not detected), background sequences any accepted
 * assay cross-reacted with (total_background = OR of the pool's background bitsets; may be NULL).  The JSON form
 * leaves the final "background matches" array unclosed when it is not empty, as the reference does (main.cpp:1235-1259). */
int64_t pcr_format_footer(const pcr_output *o, const uint8_t *target_active, const uint64_t *total_background,
	char *out, uint64_t cap);

/* ---- The design loop over all of the above (scope row f-7): what `pcramp` does between reading its FASTA files and closing
 * its output file (main.cpp:131-163, 440-443, 471-1264), one rank, one thread. */

/* The `Options` fields that loop reads (pcramp.h:83-128; the pack filters are pcr_create's, sequences, weights and deflines
 * are the loaded sets' and pcr_output's). */
typedef struct {
	uint32_t num_assay;              /* --count: stop after this many design iterations */
	uint32_t num_trial;              /* --trial: trial assays per iteration (of this rank, main.cpp:66) */
	uint32_t seed;                   /* --seed: global_seed (main.cpp:113); one rand_r draw per iteration feeds the sampler */
	int32_t  top_down_search;        /* --optimize.top-down: make_degenerate before optimize() */
	int32_t  optimize_5, optimize_3; /* shifted candidates in select_words, the trim / grow moves */
	float    target_threshold, target_search_multiplier;
	float    background_threshold, background_search_multiplier;
	float    min_target_cover, max_background_cover;
	int32_t  target_amp_min, target_amp_max, background_amp_min, background_amp_max;
	int32_t  primer_min, primer_max; /* opt.primer_range; primer_min is also Options::min_oligo_length() */
	double   max_degen;              /* -d: > 1 enables the degeneracy moves (main.cpp:82-86) */
	pcr_thermo_args thermo;
	int32_t  use_taq_mama, use_multiplex;
} pcr_design_args;

/* Runs the design loop on the loaded PCR_SET_TARGET / PCR_SET_BACKGROUND sets (weights, active flags and EOS splits as they
 * stand; the call changes them exactly as the reference changes its target_seq: detected targets become inactive, accepted
 * amplicons split their targets) and leaves the bytes of the reference's output file in the handle (pcr_design_output).
 * argc / argv: the command line the header quotes.  pool_out (optional, pool_cap entries): the accepted assays in order;
 * n_pool_out: their number.  Where the reference throws (e.g. the sampler finds no valid assay) the error code and its text
 * are returned and the output so far stays readable. */
int pcr_design(pcr_ctx *ctx, const pcr_design_args *args, const pcr_output *o, int argc, const char *const *argv,
	pcr_pair *pool_out, uint32_t pool_cap, uint32_t *n_pool_out);
/* The text of the last pcr_design call (valid until the next one or pcr_destroy). */
const char *pcr_design_output(pcr_ctx *ctx, uint64_t *len_out);

/* ---- Host-only helpers (pure CPU arithmetic of the host half of the index build; usable
 * without a GPU; exercised by the `not gpu` tests). */

/* The irregular words of one sequence (Sequence::pack's partial / EOS-adjacent emissions,
 * sequence.cpp:155-179,198-263) whose size counter is >= min_oligo_length.  Returns the count. */
int64_t pcr_host_irregular_words(const uint8_t *packed4, uint64_t len, const pcr_params *params,
	uint32_t min_oligo_length, pcr_entry *out, uint64_t cap);

/* valid_out[p] (p < len) = 1 iff the 32-base window starting at p is a regular window that
 * Sequence::pack emits under `params` (all 32 bases non-EOS; GC and degeneracy filters pass). */
int pcr_host_window_valid(const uint8_t *packed4, uint64_t len, const pcr_params *params, uint8_t *valid_out);

/* select_words' candidate list (select_words.cpp:22-85): oligo words incl. 5'/3' slot shifts and
 * their floors unsigned(size*threshold).  Returns the count (may exceed cap). */
int64_t pcr_host_candidates(const pcr_pair *pairs, uint32_t n_pairs, int optimize_5, int optimize_3,
	float threshold, pcr_word128 *words_out, uint32_t *floors_out, uint64_t cap);

/* The seed scan's filter for one candidate orientation (host model of what pcr_select_words
 * uploads; no reference counterpart -- the reference counts every window, select_words.cpp:100-117).
 * `oligo` is matched as given against 32-base windows (slot k of the word <-> base k of the
 * window); a window reaches `floor` matching slots only if, for some returned seed i, its bases
 * off[i] .. off[i]+q[i]-1 spell codes[i] (2 bits per base, A,C,G,T = 0..3, first base in the low
 * bits).  Returns the seed count (may exceed cap), or -1 if the orientation cannot be seeded
 * (the bit-sliced scan handles it). */
int64_t pcr_host_orientation_seeds(const pcr_word128 *oligo, uint32_t floor, uint32_t *codes, uint8_t *q,
	uint8_t *off, uint64_t cap);

/* The trial words of one local-search move of optimize_pcr.cpp for `oligo`, in the reference's order and
 * after its degeneracy / length gates (increase_degeneracy :17-19,:54-76; decrease_degeneracy :232-247;
 * trim5/trim3 :391-399; grow5/grow3 :671-673,:709-713), before is_valid (pcr_thermo) and the coverage
 * evaluation (pcr_move_coverage).  move: PCR_MOVE_*.  Host only.  Returns the count (may exceed cap). */
enum { PCR_MOVE_INCREASE_DEGENERACY = 0, PCR_MOVE_DECREASE_DEGENERACY = 1, PCR_MOVE_TRIM5 = 2, PCR_MOVE_TRIM3 = 3,
       PCR_MOVE_GROW5 = 4, PCR_MOVE_GROW3 = 5 };
int64_t pcr_host_move_trials(const pcr_word128 *oligo, int move, double max_degen, int primer_min, int primer_max,
	pcr_word128 *trials_out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PCRAMP_HIP_H */
