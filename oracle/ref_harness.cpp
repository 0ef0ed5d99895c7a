// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product.
//
// A thin C API around the *real* reference classes, compiled from the sources where they
// lie under /root/reference (see oracle/Makefile, target `ref`).  Nothing from the reference
// is copied here: this file only #includes its headers and calls its public functions, so
// that
//   (1) the CPU restatement in oracle/pcr_oracle.cpp can be pinned against the reference
//       itself, and golden vectors can be generated (oracle/make_golden.py), and
//   (2) bench.py can time the reference's own CPU path as `cpu_baseline.kind = "reference"`.
//
// Output: oracle/_ref/libpcramp_ref.so (git-ignored; travels to the GPU box like any built .so).
//
// Reference entry points exercised (file:line in /root/reference):
//   Word::operator&            word.cpp:68      Word::size            word.cpp:199
//   Word::complement/center    word.h:140,392   Word::begin/next      word.h:525,570
//   taq_mama_correction        word.cpp:249     Sequence::pack        sequence.cpp:92
//   select_words               select_words.cpp:8
//   PCR::find_target_match     pcr_assay.cpp:544
//   PCR::collect_target_candidates / update_target_candidates / compute_target_coverage
//                              assay.h:401,436,454 (-> pcr_assay.cpp:12,271; optimize.cpp:209)
//   Sequence::split_sequence   sequence.h:228
//   SO::SeqOverlap             seq_overlap.h / seq_overlap.cpp:347
//   NucCruc                    nuc_cruc.h:414-1450

#include <cstdint>
#include <cstring>
#include <string>
#include <sstream>
#include <vector>
#include <deque>

#include <iostream>
#include <ostream>
#include <math.h>
#include <unordered_map>
#include <unordered_set>
#include <list>
#include <set>
#include <algorithm>

// The candidate-amplicon count of PCR::collect_background_candidates is private; the harness
// needs it only to refuse the reference's odd-count out-of-bounds read (background_match.cpp:122).
#define private public
#include "assay.h"        // reference header (pulls pcramp.h, sequence.h, word.h, nuc_cruc.h)
#undef private
#include "seq_overlap.h"  // reference header

using namespace std;

// The reference declares this in main.cpp only; restated here (5 lines of bookkeeping, the
// arithmetic lives in the reference objects we link).
static inline void words_of(const Word &w, uint64_t out[2])
{
	unsigned char buf[16];
	w.mpi_pack(buf);
	memcpy(out, buf, 16);
}

static inline Word word_from(const uint64_t in[2])
{
	Word w;
	unsigned char buf[16];
	memcpy(buf, in, 16);
	w.mpi_unpack(buf);
	return w;
}

struct RefEntry
{
	uint64_t w[2];
	int32_t loc;
	uint32_t index;
	uint32_t strand; // 1 = plus, 2 = minus
	uint32_t pad;
};

struct RefSession
{
	deque<Sequence> target_seq;
	MULTIMAP<Word, WordMatch> target_db;
	vector<Word> target_keys;
	Options opt;
	string last_error;
};

extern "C" {

// ---------------------------------------------------------------- Word primitives
unsigned ref_word_and(const uint64_t a[2], const uint64_t b[2])
{
	return word_from(a) & word_from(b);
}

unsigned ref_word_size(const uint64_t a[2])
{
	return word_from(a).size();
}

int ref_word_start(const uint64_t a[2]) { return word_from(a).start(); }
int ref_word_stop(const uint64_t a[2]) { return word_from(a).stop(); }
double ref_word_degeneracy(const uint64_t a[2]) { return word_from(a).degeneracy(); }

int ref_word_from_string(const char *s, uint64_t out[2])
{
	try{
		Word w;
		w = string(s);
		words_of(w, out);
		return 0;
	}
	catch(...){ return -1; }
}

void ref_word_center(uint64_t w[2])
{
	Word x = word_from(w);
	x.center();
	words_of(x, w);
}

void ref_word_complement(const uint64_t in[2], uint64_t out[2])
{
	words_of(word_from(in).complement(), out);
}

void ref_word_shift_left(uint64_t w[2])
{
	Word x = word_from(w);
	x.shift_left();
	words_of(x, w);
}

void ref_word_shift_right(uint64_t w[2])
{
	Word x = word_from(w);
	x.shift_right();
	words_of(x, w);
}

// Expansion order of a degenerate word (Word::begin / Word::next).  Returns the count.
int ref_word_expand(const uint64_t in[2], uint64_t *out, int cap)
{
	const Word w = word_from(in);
	Word it = w.begin();
	int n = 0;
	do{
		if(n < cap){
			words_of(it, out + 2*n);
		}
		++n;
	} while( w.next(it) );
	return n;
}

float ref_taq_mama(unsigned p1, unsigned p2, unsigned t1, unsigned t2)
{
	return taq_mama_correction( make_pair( (unsigned char)p1, (unsigned char)p2 ),
		make_pair( (unsigned char)t1, (unsigned char)t2 ) );
}

// ---------------------------------------------------------------- Sequence::pack
// `seq` is IUPAC text; '-' encodes Base::EOS (a split / multi-record pad).
long ref_pack(const char *seq, unsigned index, unsigned degen_thr, float min_gc, float max_gc,
	unsigned min_len, RefEntry *out, long cap)
{
	try{
		Sequence s;
		s = string(seq);
		MULTIMAP<Word, WordMatch> db;
		s.pack(db, index, degen_thr, min_gc, max_gc, min_len);
		long n = 0;
		if( !db.empty() ){
			for(MULTIMAP<Word, WordMatch>::const_iterator i = db.begin();i != db.end();++i){
				if(n < cap){
					words_of(i->first, out[n].w);
					out[n].loc = i->second.loc;
					out[n].index = i->second.index;
					out[n].strand = (uint32_t)i->second.s;
					out[n].pad = 0;
				}
				++n;
			}
		}
		return n;
	}
	catch(const char *e){ return -1; }
	catch(...){ return -2; }
}

// ---------------------------------------------------------------- session: select_words + amplicon screen
RefSession *ref_session_create()
{
	return new RefSession();
}

void ref_session_destroy(RefSession *s)
{
	delete s;
}

const char *ref_session_error(RefSession *s) { return s->last_error.c_str(); }

int ref_session_add_target(RefSession *s, const char *seq, float weight, int active)
{
	try{
		s->target_seq.push_back( Sequence() );
		Sequence &r = s->target_seq.back();
		r = string(seq);
		r.weight(weight);
		r.active(active != 0);
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

int ref_session_set_active(RefSession *s, unsigned idx, int active)
{
	if(idx >= s->target_seq.size()) return -1;
	s->target_seq[idx].active(active != 0);
	return 0;
}

int ref_session_split(RefSession *s, unsigned idx, unsigned pos)
{
	if(idx >= s->target_seq.size()) return -1;
	if(pos >= s->target_seq[idx].length()) return -1;
	s->target_seq[idx].split_sequence(pos);
	return 0;
}

void ref_session_set_options(RefSession *s,
	float target_threshold, float search_multiplier,
	int amp_min, int amp_max, int use_taq_mama,
	unsigned pack_max_degen, float pack_min_gc, float pack_max_gc,
	int min_primer, int optimize_5, int optimize_3)
{
	Options &o = s->opt;
	o.target_threshold = target_threshold;
	o.target_search_multiplier = search_multiplier;
	o.target_amplicon_range = make_pair(amp_min, amp_max);
	o.use_taq_mama = (use_taq_mama != 0);
	o.pack_max_degen = pack_max_degen;
	o.pack_min_gc = pack_min_gc;
	o.pack_max_gc = pack_max_gc;
	o.primer_range.first = min_primer;
	o.optimize_5 = (optimize_5 != 0);
	o.optimize_3 = (optimize_3 != 0);
}

// The per-iteration word-DB build of main.cpp:644-691 with the given trial assays
// (pairs = n x {F[2], R[2]} 64-bit words).  `threshold` < 0 selects the reference's own
// target_threshold*target_search_multiplier (main.cpp:669); `min_len_override` >= 0 replaces
// opt.min_oligo_length() (backgrounds use 0.9x, main.cpp:595).
long ref_session_select(RefSession *s, const uint64_t *pairs, unsigned n_pairs,
	float threshold, int min_len_override)
{
	try{
		vector<PCR> trial(n_pairs);
		for(unsigned i = 0;i < n_pairs;++i){
			trial[i].oligo( FORWARD, word_from(pairs + 4*i) );
			trial[i].oligo( REVERSE, word_from(pairs + 4*i + 2) );
		}

		const Options &opt = s->opt;
		const float thr = (threshold < 0.0f) ?
			opt.target_threshold*opt.target_search_multiplier : threshold;
		const unsigned min_len = (min_len_override >= 0) ?
			(unsigned)min_len_override : (unsigned)opt.min_oligo_length();

		s->target_db = MULTIMAP<Word, WordMatch>();

		const unsigned n = s->target_seq.size();
		for(unsigned i = 0;i < n;++i){
			if( !s->target_seq[i].active() ){
				continue;
			}
			MULTIMAP<Word, WordMatch> local_db;
			s->target_seq[i].pack(local_db, i, opt.pack_max_degen, opt.pack_min_gc,
				opt.pack_max_gc, min_len);
			select_words(s->target_db, local_db, trial, opt.optimize_5, opt.optimize_3, thr);
		}
		s->target_db.sort();
		s->target_keys = keys(s->target_db);
		return (long)s->target_db.size();
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

long ref_session_db_entries(RefSession *s, RefEntry *out, long cap)
{
	long n = 0;
	if( s->target_db.empty() ) return 0;
	for(MULTIMAP<Word, WordMatch>::const_iterator i = s->target_db.begin();i != s->target_db.end();++i){
		if(n < cap){
			words_of(i->first, out[n].w);
			out[n].loc = i->second.loc;
			out[n].index = i->second.index;
			out[n].strand = (uint32_t)i->second.s;
			out[n].pad = 0;
		}
		++n;
	}
	return n;
}

// PCR::find_target_match (pcr_assay.cpp:544): bits_out[i] = 0/1 per target.
int ref_session_target_match(RefSession *s, const uint64_t pair[4], unsigned char *bits_out)
{
	try{
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		BitSet m;
		p.find_target_match(m, s->target_keys, s->target_db, s->target_seq, s->opt);
		for(size_t i = 0;i < s->target_seq.size();++i){
			bits_out[i] = ( (i < m.size()) && m[i] ) ? 1 : 0;
		}
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// The coverage evaluation of optimize() (optimize.cpp:61-74): collect with the search
// threshold, update identities, sum weights.
float ref_session_target_coverage(RefSession *s, const uint64_t pair[4])
{
	try{
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		p.collect_target_candidates(s->target_keys, s->target_db, s->target_seq, s->opt);
		p.update_target_candidates(s->target_keys, s->opt.use_taq_mama);
		return p.compute_target_coverage(s->opt.target_threshold);
	}
	catch(const char *e){ s->last_error = e; return -1.0f; }
	catch(...){ s->last_error = "unknown"; return -2.0f; }
}

// A local-search move as optimize_pcr.cpp evaluates it (:77-93 and the same lines of the other moves):
// candidates of the base assay (collected by optimize(), optimize.cpp:61-70), identity table of the
// modified oligo recomputed for the trial word, compute_target_coverage.
int ref_session_move_coverage(RefSession *s, const uint64_t base[4], int side, const uint64_t *variants,
	unsigned n_variants, float *cov_out)
{
	try{
		PCR p;
		p.oligo( FORWARD, word_from(base) );
		p.oligo( REVERSE, word_from(base + 2) );
		p.collect_target_candidates(s->target_keys, s->target_db, s->target_seq, s->opt);
		p.update_target_candidates(s->target_keys, s->opt.use_taq_mama);
		for(unsigned v = 0;v < n_variants;++v){
			const Word trial = word_from(variants + 2*v);
			update_identity( (side == 0) ? p.target_f_identity : p.target_r_identity, trial, s->target_keys, s->opt.use_taq_mama );
			cov_out[v] = p.compute_target_coverage(s->opt.target_threshold);
		}
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// One complete local-search move the way optimize() drives it (optimize.cpp:61-79,126-130): candidates and
// identity tables of the base assay on the target session and the optional background session, base score
// as the score threshold, then optimization_move() -- the reference's own move function -- with a fresh
// NucCruc.  Non-multiplex (empty pool, empty multiplex keys).
struct RefMoveOptions {
	int degen; int primer_min, primer_max;
	float salt, primer_strand, tm_min, tm_max, max_hairpin;
	float bg_threshold, bg_multiplier; int bg_amp_min, bg_amp_max;
};

static void prefill(NucCruc &melt);

int ref_optimization_move(RefSession *t, RefSession *b, const uint64_t pair[4], int move, int side,
	const RefMoveOptions *mo, uint64_t out_word[2], float out_score[3], float base_score_out[2])
{
	try{
		Options opt = t->opt;
		opt.degen = mo->degen;
		opt.primer_range = make_pair(mo->primer_min, mo->primer_max);
		opt.salt = mo->salt; opt.primer_strand = mo->primer_strand;
		opt.primer_tm_range = make_pair(mo->tm_min, mo->tm_max);
		opt.max_hairpin = mo->max_hairpin;
		opt.background_threshold = mo->bg_threshold;
		opt.background_search_multiplier = mo->bg_multiplier;
		opt.background_amplicon_range = make_pair(mo->bg_amp_min, mo->bg_amp_max);
		opt.use_multiplex = false;
		const vector<Word> no_keys;
		const MULTIMAP<Word, WordMatch> no_db;
		const deque<Sequence> no_seq;
		const vector<Word> &bkeys = b ? b->target_keys : no_keys;
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		p.collect_target_candidates(t->target_keys, t->target_db, t->target_seq, opt);
		p.collect_background_candidates(bkeys, b ? b->target_db : no_db, b ? b->target_seq : no_seq, opt);
		p.update_target_candidates(t->target_keys, opt.use_taq_mama);
		p.update_background_candidates(bkeys, opt.use_taq_mama);
		Score base;
		base.target_coverage = p.compute_target_coverage(opt.target_threshold);
		base.background_coverage = p.compute_background_coverage(opt.background_threshold);
		if(base_score_out){ base_score_out[0] = base.target_coverage; base_score_out[1] = base.background_coverage; }
		NucCruc melt;
		prefill(melt);
		melt.fast_alignment(true);
		melt.salt(opt.salt);
		const deque<PCR> pool;
		const Move mv[6] = { IncreaseDegeneracy, DecreaseDegeneracy, Trim5, Trim3, Grow5, Grow3 };
		if(move < 0 || move > 5) throw "unknown move";
		const std::pair<Word, Score> r = optimization_move(mv[move], (side == 0) ? FORWARD : REVERSE, p,
			t->target_keys, opt.target_threshold, bkeys, opt.background_threshold, no_keys, base, melt, pool, opt);
		unsigned char buf[16];
		r.first.mpi_pack(buf);
		memcpy(out_word, buf, 16);
		out_score[0] = r.second.target_coverage; out_score[1] = r.second.background_coverage; out_score[2] = r.second.oligo_overlap;
		return 0;
	}
	catch(const char *e){ t->last_error = e; return -1; }
	catch(...){ t->last_error = "unknown"; return -2; }
}

// optimize() itself (optimize.cpp:14-207), non-multiplex.
int ref_optimize(RefSession *t, RefSession *b, uint64_t pair_inout[4], const int *moves, int n_moves,
	const RefMoveOptions *mo, float out_score[3], int *iterations_out)
{
	try{
		Options opt = t->opt;
		opt.degen = mo->degen;
		opt.primer_range = make_pair(mo->primer_min, mo->primer_max);
		opt.salt = mo->salt; opt.primer_strand = mo->primer_strand;
		opt.primer_tm_range = make_pair(mo->tm_min, mo->tm_max);
		opt.max_hairpin = mo->max_hairpin;
		opt.background_threshold = mo->bg_threshold;
		opt.background_search_multiplier = mo->bg_multiplier;
		opt.background_amplicon_range = make_pair(mo->bg_amp_min, mo->bg_amp_max);
		opt.use_multiplex = false;
		opt.output_filter = Options::SILENT;
		const vector<Word> no_keys;
		const MULTIMAP<Word, WordMatch> no_db;
		const deque<Sequence> no_seq;
		const Move mv[6] = { IncreaseDegeneracy, DecreaseDegeneracy, Trim5, Trim3, Grow5, Grow3 };
		vector<Move> ml;
		for(int i = 0;i < n_moves;++i){ if(moves[i] < 0 || moves[i] > 5) throw "unknown move"; ml.push_back(mv[moves[i]]); }
		PCR p;
		p.oligo( FORWARD, word_from(pair_inout) );
		p.oligo( REVERSE, word_from(pair_inout + 2) );
		const deque<PCR> pool;
		std::ostringstream sink;
		const Score sc = optimize(p, ml, t->target_keys, t->target_db, t->target_seq,
			b ? b->target_keys : no_keys, b ? b->target_db : no_db, b ? b->target_seq : no_seq,
			no_keys, no_db, no_seq, pool, opt, sink);
		words_of(p.oligo(FORWARD), pair_inout);
		words_of(p.oligo(REVERSE), pair_inout + 2);
		out_score[0] = sc.target_coverage; out_score[1] = sc.background_coverage; out_score[2] = sc.oligo_overlap;
		if(iterations_out) *iterations_out = -1;
		return 0;
	}
	catch(const char *e){ t->last_error = e; return -1; }
	catch(...){ t->last_error = "unknown"; return -2; }
}

// make_degenerate itself (optimize.cpp:356-398), with one NucCruc as main.cpp:702-716 gives it.
int ref_make_degenerate(RefSession *t, uint64_t pair_inout[4], const RefMoveOptions *mo, float max_dimer, int *valid_out)
{
	try{
		Options opt = t->opt;
		opt.degen = mo->degen;
		opt.primer_range = make_pair(mo->primer_min, mo->primer_max);
		opt.salt = mo->salt; opt.primer_strand = mo->primer_strand;
		opt.primer_tm_range = make_pair(mo->tm_min, mo->tm_max);
		opt.max_hairpin = mo->max_hairpin;
		opt.max_dimer = max_dimer;
		opt.output_filter = Options::SILENT;
		PCR p;
		p.oligo( FORWARD, word_from(pair_inout) );
		p.oligo( REVERSE, word_from(pair_inout + 2) );
		NucCruc melt;
		prefill(melt);
		melt.salt(opt.salt);
		std::ostringstream sink;
		const bool ok = make_degenerate(p, t->target_keys, t->target_db, t->target_seq, melt, opt, sink);
		words_of(p.oligo(FORWARD), pair_inout);
		words_of(p.oligo(REVERSE), pair_inout + 2);
		if(valid_out) *valid_out = ok ? 1 : 0;
		return 0;
	}
	catch(const char *e){ t->last_error = e; return -1; }
	catch(...){ t->last_error = "unknown"; return -2; }
}

// One move with opt.use_multiplex: candidates / identity tables / Score of the base assay as optimize() builds
// them (optimize.cpp:61-97), then the reference's own optimization_move().
int ref_optimization_move_multiplex(RefSession *t, RefSession *b, RefSession *amplicons, const uint64_t *pool_words, unsigned n_pool,
	const uint64_t pair[4], int move, int side, const RefMoveOptions *mo, uint64_t out_word[2], float out_score[3], float base_score_out[3])
{
	try{
		Options opt = t->opt;
		opt.degen = mo->degen;
		opt.primer_range = make_pair(mo->primer_min, mo->primer_max);
		opt.salt = mo->salt; opt.primer_strand = mo->primer_strand;
		opt.primer_tm_range = make_pair(mo->tm_min, mo->tm_max);
		opt.max_hairpin = mo->max_hairpin;
		opt.background_threshold = mo->bg_threshold;
		opt.background_search_multiplier = mo->bg_multiplier;
		opt.background_amplicon_range = make_pair(mo->bg_amp_min, mo->bg_amp_max);
		opt.use_multiplex = true;
		const vector<Word> no_keys;
		const MULTIMAP<Word, WordMatch> no_db;
		const deque<Sequence> no_seq;
		const vector<Word> &bkeys = b ? b->target_keys : no_keys;
		MULTIMAP<Word, WordMatch> mdb;
		deque<Sequence> mseq;
		for(deque<Sequence>::const_iterator i = amplicons->target_seq.begin();i != amplicons->target_seq.end();++i){
			i->pack(mdb, mseq.size(), amplicons->opt.pack_max_degen, 0.0, 1.0, amplicons->opt.min_oligo_length());
			mseq.push_back(*i);
		}
		mdb.sort();
		const vector<Word> mkeys = keys(mdb);
		deque<PCR> pool(n_pool);
		for(unsigned i = 0;i < n_pool;++i){
			pool[i].oligo( FORWARD, word_from(pool_words + 4*i) );
			pool[i].oligo( REVERSE, word_from(pool_words + 4*i + 2) );
		}
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		p.collect_target_candidates(t->target_keys, t->target_db, t->target_seq, opt);
		p.collect_background_candidates(bkeys, b ? b->target_db : no_db, b ? b->target_seq : no_seq, opt);
		p.update_target_candidates(t->target_keys, opt.use_taq_mama);
		p.update_background_candidates(bkeys, opt.use_taq_mama);
		Score base;
		base.target_coverage = p.compute_target_coverage(opt.target_threshold);
		base.background_coverage = p.compute_background_coverage(opt.background_threshold);
		p.collect_multiplex_background_candidates(mkeys, mdb, mseq, opt);
		p.update_multiplex_background_candidates(mkeys, opt.use_taq_mama);
		base.background_coverage += p.compute_multiplex_background_coverage(opt.background_threshold);
		base.oligo_overlap = p.compute_oligo_overlap(pool);
		if(base_score_out){ base_score_out[0] = base.target_coverage; base_score_out[1] = base.background_coverage; base_score_out[2] = base.oligo_overlap; }
		NucCruc melt;
		prefill(melt);
		melt.fast_alignment(true);
		melt.salt(opt.salt);
		const Move mv[6] = { IncreaseDegeneracy, DecreaseDegeneracy, Trim5, Trim3, Grow5, Grow3 };
		if(move < 0 || move > 5) throw "unknown move";
		const std::pair<Word, Score> r = optimization_move(mv[move], (side == 0) ? FORWARD : REVERSE, p,
			t->target_keys, opt.target_threshold, bkeys, opt.background_threshold, mkeys, base, melt, pool, opt);
		unsigned char buf[16];
		r.first.mpi_pack(buf);
		memcpy(out_word, buf, 16);
		out_score[0] = r.second.target_coverage; out_score[1] = r.second.background_coverage; out_score[2] = r.second.oligo_overlap;
		return 0;
	}
	catch(const char *e){ t->last_error = e; return -1; }
	catch(...){ t->last_error = "unknown"; return -2; }
}

// optimize() with opt.use_multiplex: multiplex background DB = pack of the `amplicons` session's sequences
// (main.cpp:989-1001), pool = the assays designed so far.
int ref_optimize_multiplex(RefSession *t, RefSession *b, RefSession *amplicons, const uint64_t *pool_words, unsigned n_pool,
	uint64_t pair_inout[4], const int *moves, int n_moves, const RefMoveOptions *mo, float out_score[3])
{
	try{
		Options opt = t->opt;
		opt.degen = mo->degen;
		opt.primer_range = make_pair(mo->primer_min, mo->primer_max);
		opt.salt = mo->salt; opt.primer_strand = mo->primer_strand;
		opt.primer_tm_range = make_pair(mo->tm_min, mo->tm_max);
		opt.max_hairpin = mo->max_hairpin;
		opt.background_threshold = mo->bg_threshold;
		opt.background_search_multiplier = mo->bg_multiplier;
		opt.background_amplicon_range = make_pair(mo->bg_amp_min, mo->bg_amp_max);
		opt.use_multiplex = true;
		opt.output_filter = Options::SILENT;
		const vector<Word> no_keys;
		const MULTIMAP<Word, WordMatch> no_db;
		const deque<Sequence> no_seq;
		MULTIMAP<Word, WordMatch> mdb;
		deque<Sequence> mseq;
		for(deque<Sequence>::const_iterator i = amplicons->target_seq.begin();i != amplicons->target_seq.end();++i){
			i->pack(mdb, mseq.size(), amplicons->opt.pack_max_degen, 0.0, 1.0, amplicons->opt.min_oligo_length());
			mseq.push_back(*i);
		}
		mdb.sort();
		const vector<Word> mkeys = keys(mdb);
		const Move mv[6] = { IncreaseDegeneracy, DecreaseDegeneracy, Trim5, Trim3, Grow5, Grow3 };
		vector<Move> ml;
		for(int i = 0;i < n_moves;++i){ if(moves[i] < 0 || moves[i] > 5) throw "unknown move"; ml.push_back(mv[moves[i]]); }
		PCR p;
		p.oligo( FORWARD, word_from(pair_inout) );
		p.oligo( REVERSE, word_from(pair_inout + 2) );
		deque<PCR> pool(n_pool);
		for(unsigned i = 0;i < n_pool;++i){
			pool[i].oligo( FORWARD, word_from(pool_words + 4*i) );
			pool[i].oligo( REVERSE, word_from(pool_words + 4*i + 2) );
		}
		std::ostringstream sink;
		const Score sc = optimize(p, ml, t->target_keys, t->target_db, t->target_seq,
			b ? b->target_keys : no_keys, b ? b->target_db : no_db, b ? b->target_seq : no_seq,
			mkeys, mdb, mseq, pool, opt, sink);
		words_of(p.oligo(FORWARD), pair_inout);
		words_of(p.oligo(REVERSE), pair_inout + 2);
		out_score[0] = sc.target_coverage; out_score[1] = sc.background_coverage; out_score[2] = sc.oligo_overlap;
		return 0;
	}
	catch(const char *e){ t->last_error = e; return -1; }
	catch(...){ t->last_error = "unknown"; return -2; }
}

// ---------------------------------------------------------------- Smith-Waterman (seq_overlap)
// One 8-lane call exactly as background_match.cpp drives it: queries/targets are arrays of
// SO_LEN 64-bit-pair Words (slot i of each).  Outputs per lane: score, query range, target
// range, last-two-aligned target bases.
int ref_sw_align_words(const uint64_t *query_words, const uint64_t *target_words,
	int16_t *score, int32_t *qrange, int32_t *trange, uint8_t *last_two)
{
	try{
		SO::SeqOverlap align(SO::SeqOverlap::SmithWaterman, true /*is_na*/);
		for(unsigned i = 0;i < SO_LEN;++i){
			align.pack_query_slots( (unsigned char)(1u << i), word_from(query_words + 2*i) );
			align.pack_target_slots( (unsigned char)(1u << i), word_from(target_words + 2*i) );
		}
		align.align();
		for(unsigned i = 0;i < SO_LEN;++i){
			score[i] = align.score(i);
			const pair<int,int> q = align.alignment_range_query(i);
			const pair<int,int> t = align.alignment_range_target(i);
			qrange[2*i] = q.first; qrange[2*i + 1] = q.second;
			trange[2*i] = t.first; trange[2*i + 1] = t.second;
			const pair<unsigned char, unsigned char> l2 = align.target_last_two_aligned(i);
			last_two[2*i] = l2.first; last_two[2*i + 1] = l2.second;
		}
		return 0;
	}
	catch(const char *e){ return -1; }
	catch(...){ return -2; }
}


// PCR::find_background_match (background_match.cpp:7) with the session's sequences as the
// background set.  Returns the candidate amplicon count, or -3 WITHOUT evaluating when that
// count is odd and smaller than the number of sequences: the reference then reads lanes 4-7 of a
// previous iteration and indexes past its amplicon deque (background_match.cpp:122).
long ref_session_background_match(RefSession *s, const uint64_t pair[4], float bg_threshold, float bg_multiplier,
	int amp_min, int amp_max, int use_taq_mama, unsigned char *bits_out)
{
	try{
		Options opt = s->opt;
		opt.background_threshold = bg_threshold;
		opt.background_search_multiplier = bg_multiplier;
		opt.background_amplicon_range = make_pair(amp_min, amp_max);
		opt.use_taq_mama = (use_taq_mama != 0);
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		p.collect_background_candidates(s->target_keys, s->target_db, s->target_seq, opt);
		const long n_amp = (long)p.background_amplicons.size();
		const size_t n = s->target_seq.size();
		memset(bits_out, 0, n);
		if( (n_amp & 1) && ( (size_t)n_amp < n ) ){
			return -3;
		}
		BitSet m(n, false);
		p.find_background_match(m, s->target_keys, s->target_db, s->target_seq, opt, cerr);
		for(size_t i = 0;i < n;++i){
			bits_out[i] = m[i] ? 1 : 0;
		}
		return n_amp;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

int ref_session_multiplex_match(RefSession *s, const uint64_t pair[4], float bg_threshold, int use_taq_mama,
	unsigned char *bits_out)
{
	try{
		Options opt = s->opt;
		opt.background_threshold = bg_threshold;
		opt.use_taq_mama = (use_taq_mama != 0);
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		const size_t n = s->target_seq.size();
		BitSet m(n, false);
		p.find_multiplex_background_match(m, s->target_seq, opt, cerr);
		for(size_t i = 0;i < n;++i){
			bits_out[i] = m[i] ? 1 : 0;
		}
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// ---------------------------------------------------------------- NucCruc thermodynamics
// out[0..3] = tm, dH, dS, dG(37C) of the perfect-match duplex; out[4] hairpin Tm;
// out[5] homodimer Tm.
int ref_thermo(const char *seq, float salt, float strand, float *out)
{
	try{
		NucCruc melt;
		melt.salt(salt);
		melt.strand(strand);
		const string s(seq);
		out[0] = melt.tm_pm_duplex(s);
		out[1] = melt.delta_H();
		out[2] = melt.delta_S();
		out[3] = melt.delta_G();
		melt.set_query(s);
		out[4] = melt.approximate_tm_hairpin();
		out[5] = melt.approximate_tm_homodimer();
		return 0;
	}
	catch(const char *e){ return -1; }
	catch(...){ return -2; }
}

float ref_heterodimer(const char *a, const char *b, float salt, float strand_a, float strand_b)
{
	try{
		NucCruc melt;
		melt.salt(salt);
		melt.strand(strand_a, strand_b);
		melt.set_query( string(a) );
		melt.set_target( string(b) );
		return melt.approximate_tm_heterodimer();
	}
	catch(...){ return -1.0f; }
}

// ---------------------------------------------------------------- NucCruc, deterministic form
// NucCruc::trace_back reads m_q[query_len] (one element past the query, nuc_cruc.cpp:1383 with
// last_i == 0) whenever a path runs into DP row 0; CircleBuffer::operator[] then returns whatever
// an earlier, longer query left in the ring (or uninitialised memory).  To make the reference a
// function of its inputs the harness first fills both rings with the dangling-end code E, which
// pairs with nothing, so the stray column is always trimmed (nuc_cruc.cpp:868-884).
static void prefill(NucCruc &melt)
{
	for(unsigned i = 0;i < MAX_SEQUENCE_LENGTH;++i){   // straight into the rings: push_back_query() refuses E
		melt.query.push_back(BASE::E);
		melt.target.push_back(BASE::E);
	}
	melt.clear();
}

// out: [0] tm_pm_duplex, [1] dH, [2] dS, [3] dG; [4] hairpin tm, [5] dH, [6] dS; [7] homodimer tm, [8] dH, [9] dS
int ref_thermo_full(const char *seq, float salt, float strand, float *out)
{
	try{
		NucCruc melt;
		prefill(melt);
		melt.salt(salt);
		melt.strand(strand);
		const string s(seq);
		out[0] = melt.tm_pm_duplex(s); out[1] = melt.delta_H(); out[2] = melt.delta_S(); out[3] = melt.delta_G();
		melt.set_query(s);
		out[4] = melt.approximate_tm_hairpin(); out[5] = melt.delta_H(); out[6] = melt.delta_S();
		out[7] = melt.approximate_tm_homodimer(); out[8] = melt.delta_H(); out[9] = melt.delta_S();
		return 0;
	}
	catch(const char *e){ return -1; }
	catch(...){ return -2; }
}

// out: tm, dH, dS
int ref_heterodimer_full(const char *a, const char *b, float salt, float strand_a, float strand_b, float *out)
{
	try{
		NucCruc melt;
		prefill(melt);
		melt.salt(salt);
		melt.strand(strand_a, strand_b);
		melt.set_query( string(a) );
		melt.set_target( string(b) );
		out[0] = melt.approximate_tm_heterodimer(); out[1] = melt.delta_H(); out[2] = melt.delta_S();
		return 0;
	}
	catch(...){ return -1; }
}

static void thermo_options(Options &opt, float primer_strand, float tm_min, float tm_max, float max_hairpin, float max_dimer)
{
	opt.primer_strand = primer_strand;
	opt.primer_tm_range = make_pair(tm_min, tm_max);
	opt.max_hairpin = max_hairpin;
	opt.max_dimer = max_dimer;
}

// PCR::is_valid (valid_pcr.cpp:5-45)
int ref_is_valid(const uint64_t word[2], float salt, float primer_strand, float tm_min, float tm_max,
	float max_hairpin, float max_dimer, int check_homo_dimer)
{
	try{
		NucCruc melt;
		prefill(melt);
		melt.salt(salt);
		Options opt;
		thermo_options(opt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer);
		PCR p;
		const Word w = word_from(word);
		return p.is_valid(FORWARD, w, melt, opt, check_homo_dimer != 0) ? 1 : 0;
	}
	catch(...){ return -1; }
}

// PCR::max_dimer_tm (pcr_assay.cpp:232-269)
float ref_max_dimer_tm(const uint64_t pair[4], float salt, float primer_strand)
{
	try{
		NucCruc melt;
		prefill(melt);
		melt.salt(salt);
		Options opt;
		opt.primer_strand = primer_strand;
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		return p.max_dimer_tm(melt, opt);
	}
	catch(...){ return -1.0f; }
}

// PCR::multiplex_compatible (pcr_assay.cpp:815-852): this->assay = a, argument = b
int ref_multiplex_compatible(const uint64_t a[4], const uint64_t b[4], float salt, float primer_strand, float max_dimer)
{
	try{
		NucCruc melt;
		prefill(melt);
		melt.salt(salt);
		Options opt;
		opt.primer_strand = primer_strand;
		opt.max_dimer = max_dimer;
		PCR pa, pb;
		pa.oligo( FORWARD, word_from(a) ); pa.oligo( REVERSE, word_from(a + 2) );
		pb.oligo( FORWARD, word_from(b) ); pb.oligo( REVERSE, word_from(b + 2) );
		return pa.multiplex_compatible(melt, opt, pb) ? 1 : 0;
	}
	catch(...){ return -1; }
}

// PCR::random_assay (pcr_assay.cpp:580-734) for n_trials fresh assays on one running seed -- the body of
// the one-thread sampling loop, main.cpp:544-550.  One NucCruc serves all calls, as there; it is pre-filled
// once (see prefill), so later calls see whatever earlier queries left in the rings, as in the program.
struct RefSamplerOptions {
	int primer_min, primer_max, amp_min, amp_max;
	double max_degen;
	float salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer;
};

unsigned ref_rand_r(unsigned *seed) { return (unsigned)rand_r(seed); }

int ref_random_assays(RefSession *s, unsigned *seed, unsigned n_trials, const RefSamplerOptions *o, uint64_t *pairs_out)
{
	try{
		Options opt = s->opt;
		opt.primer_range = make_pair(o->primer_min, o->primer_max);
		opt.target_amplicon_range = make_pair(o->amp_min, o->amp_max);
		opt.degen = o->max_degen;
		opt.salt = o->salt;
		thermo_options(opt, o->primer_strand, o->tm_min, o->tm_max, o->max_hairpin, o->max_dimer);
		NucCruc melt;
		prefill(melt);
		melt.salt(opt.salt);
		ostringstream sink;
		for(unsigned t = 0;t < n_trials;++t){
			PCR p;
			p.random_assay(s->target_seq, melt, opt, *seed, sink);
			words_of(p.oligo(FORWARD), pairs_out + 4*t);
			words_of(p.oligo(REVERSE), pairs_out + 4*t + 2);
		}
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// Word::max_overlap (word.h:38-91) and PCR::compute_oligo_overlap (pcr_assay.cpp:736-754)
float ref_word_max_overlap(const uint64_t a[2], const uint64_t b[2])
{
	return word_from(a).max_overlap( word_from(b) );
}

float ref_oligo_overlap(const uint64_t assay[4], const uint64_t *pool, unsigned n_pool)
{
	PCR p;
	p.oligo( FORWARD, word_from(assay) );
	p.oligo( REVERSE, word_from(assay + 2) );
	deque<PCR> q(n_pool);
	for(unsigned i = 0;i < n_pool;++i){
		q[i].oligo( FORWARD, word_from(pool + 4*i) );
		q[i].oligo( REVERSE, word_from(pool + 4*i + 2) );
	}
	return p.compute_oligo_overlap(q);
}

// PCR::write / PCR::write_json (assay.h:288-375), all four forms: with_pool = 0 selects the pool-less overloads.
// Returns the length of the text; at most cap bytes are stored.
long ref_format_oligos(const uint64_t assay[4], const uint64_t *pool, unsigned n_pool, int json, int with_pool, char *out, long cap)
{
	try{
		PCR p;
		p.oligo( FORWARD, word_from(assay) );
		p.oligo( REVERSE, word_from(assay + 2) );
		deque<PCR> q(n_pool);
		for(unsigned i = 0;i < n_pool;++i){
			q[i].oligo( FORWARD, word_from(pool + 4*i) );
			q[i].oligo( REVERSE, word_from(pool + 4*i + 2) );
		}
		ostringstream ss;
		if(json){ if(with_pool) p.write_json(ss, q); else p.write_json(ss); }
		else{ if(with_pool) p.write(ss, q); else p.write(ss); }
		const string t = ss.str();
		for(long i = 0;i < cap && i < (long)t.size();++i) out[i] = t[i];
		return (long)t.size();
	}
	catch(...){ return -1; }
}

// PCR::collect_unique_amplicons (pcr_assay.cpp:756-813) over the session's DB: bounds in discovery order, the
// unique amplicon Sequences as nibbles back to back.
long ref_session_collect_amplicons(RefSession *s, const uint64_t pair[4], float threshold, int amp_min, int amp_max,
	unsigned *bounds_out, long cap_bounds, unsigned char *amp_codes_out, long cap_codes, unsigned *amp_len_out, long cap_amp,
	long *n_amp_out)
{
	try{
		PCR p;
		p.oligo( FORWARD, word_from(pair) );
		p.oligo( REVERSE, word_from(pair + 2) );
		deque<AmpliconBounds> bounds;
		const deque<Sequence> amps = p.collect_unique_amplicons(s->target_keys, s->target_db, s->target_seq, threshold,
			make_pair(amp_min, amp_max), &bounds);
		long used = 0, na = 0;
		for(deque<Sequence>::const_iterator a = amps.begin();a != amps.end();++a, ++na){
			if(na < cap_amp) amp_len_out[na] = (unsigned)a->length();
			for(size_t i = 0;i < a->length();++i, ++used){ if(used < cap_codes) amp_codes_out[used] = (*a)[i]; }
		}
		if(n_amp_out) *n_amp_out = na;
		long nb = 0;
		for(deque<AmpliconBounds>::const_iterator b = bounds.begin();b != bounds.end();++b, ++nb){
			if(nb < cap_bounds){ bounds_out[3*nb] = b->index; bounds_out[3*nb + 1] = b->begin; bounds_out[3*nb + 2] = b->end; }
		}
		return (used > cap_codes || na > cap_amp) ? -3 : nb;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// The multiplex background coverage as optimize() / the moves evaluate it (optimize.cpp:82-92,
// optimize_pcr.cpp:111-127): DB = pack of the session's sequences the way main.cpp:989-1001 packs accepted
// amplicons, candidates of the base assay, identity map of the edited oligo recomputed per trial word.
int ref_multiplex_coverage(RefSession *s, const uint64_t base[4], int side, const uint64_t *variants, unsigned n_variants,
	float background_threshold, int use_taq_mama, float *cov_out, unsigned *n_keys_out)
{
	try{
		MULTIMAP<Word, WordMatch> db;
		deque<Sequence> seqs;
		for(deque<Sequence>::const_iterator i = s->target_seq.begin();i != s->target_seq.end();++i){
			i->pack(db, seqs.size(), s->opt.pack_max_degen, 0.0, 1.0, s->opt.min_oligo_length());
			seqs.push_back(*i);
		}
		db.sort();
		const vector<Word> mkeys = keys(db);
		if(n_keys_out) *n_keys_out = (unsigned)mkeys.size();
		Options opt = s->opt;
		opt.background_threshold = background_threshold;
		opt.use_taq_mama = (use_taq_mama != 0);
		PCR p;
		p.oligo( FORWARD, word_from(base) );
		p.oligo( REVERSE, word_from(base + 2) );
		p.collect_multiplex_background_candidates(mkeys, db, seqs, opt);
		p.update_multiplex_background_candidates(mkeys, opt.use_taq_mama);
		for(unsigned v = 0;v < n_variants;++v){
			const Word trial = word_from(variants + 2*v);
			update_identity( (side == 0) ? p.multiplex_background_f_identity : p.multiplex_background_r_identity, trial, mkeys, opt.use_taq_mama );
			cov_out[v] = p.compute_multiplex_background_coverage(opt.background_threshold);
		}
		return 0;
	}
	catch(const char *e){ s->last_error = e; return -1; }
	catch(...){ s->last_error = "unknown"; return -2; }
}

// The SantaLucia parameter set as the reference initialises it (published values:
// SantaLucia & Hicks, Annu. Rev. Biophys. Biomol. Struct. 33:415-440, 2004), for
// oracle/gen_thermo_tables.py.  scalars: init_H, init_S, asymmetric_loop_dS, bulge_AT_closing_S,
// AT_closing_H, AT_closing_S, symmetry_S, SALT.
int ref_thermo_tables(float *H, float *S, float *loop_S, float *bulge_S, float *hairpin_S, float *special_H,
	float *special_S, float *scalars, float *supp, float *supp_salt, unsigned char *wc)
{
	NucCruc m;
	for(int i = 0;i < NUM_BASE_PAIR;++i){
		for(int j = 0;j < NUM_BASE_PAIR;++j){
			H[i*NUM_BASE_PAIR + j] = m.param_H[i][j];
			S[i*NUM_BASE_PAIR + j] = m.param_S[i][j];
			if(m.param_loop_terminal_H[i][j] != m.param_H[i][j] || m.param_loop_terminal_S[i][j] != m.param_S[i][j] ||
			   m.param_hairpin_terminal_H[i][j] != m.param_H[i][j] || m.param_hairpin_terminal_S[i][j] != m.param_S[i][j]){
				return -1;   // the terminal tables are documented copies of H/S (nuc_cruc_santa_lucia.cpp:594-601)
			}
		}
		wc[i] = m.watson_and_crick[i] ? 1 : 0;
	}
	for(int i = 0;i <= MAX_LOOP_LENGTH;++i){ loop_S[i] = m.param_loop_S[i]; bulge_S[i] = m.param_bulge_S[i]; hairpin_S[i] = m.param_hairpin_S[i]; }
	for(int i = 0;i < NucCruc::NUM_SPECIAL_HAIRPIN_LOOP;++i){ special_H[i] = m.param_hairpin_special_H[i]; special_S[i] = m.param_hairpin_special_S[i]; }
	scalars[0] = m.param_init_H; scalars[1] = m.param_init_S; scalars[2] = m.param_asymmetric_loop_dS;
	scalars[3] = m.param_bulge_AT_closing_S; scalars[4] = m.param_AT_closing_H; scalars[5] = m.param_AT_closing_S;
	scalars[6] = m.param_symmetry_S; scalars[7] = m.param_SALT;
	for(int i = 0;i < NucCruc::NUM_SUPP_PARAM;++i) supp[i] = m.param_supp[i];
	for(int i = 0;i < NucCruc::NUM_SALT_PARAM;++i) supp_salt[i] = m.param_supp_salt[i];
	return NucCruc::NUM_SPECIAL_HAIRPIN_LOOP;
}

} // extern "C"
